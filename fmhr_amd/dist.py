"""fmhr_amd.dist — view sharding of the HAM step over ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm) — SURVEY.md §8e.  The reference has no multi-GPU code; this is the build's data-parallel axis:

* the n views of a step are split contiguously over G ranks; every rank holds the whole mesh;
* ONE all-reduce(SUM) per step and nothing else.  F.l1_loss divides by the batch-GLOBAL 3*Pv (mesh_sfs_optim.py:289), which a
  rank only knows after an exchange; instead of exchanging the count between forward and backward (a second, blocking
  collective in the middle of a ~0.15 ms step) every rank back-propagates the photometric term with the divisor LEFT OUT and
  keeps that share of the gradient apart from the rest (the photometric gradient only flows through the rgb channels, the
  silhouette MSE only through alpha, so the split is free): bucket = [P_delta | P_albedo | R_delta | R_albedo | 8 loss slots] =
  12V+8 fp32 (2.35 MB at V = 49k: latency-bound on xGMI).  After the all-reduce, g = P / (3 * n_valid_global) + R
  (fmhr_ham_combine), n_valid_global being the summed count slot.  Every entry is the rank's share: the view terms of its
  views and the view-independent terms (regularisers) of its VERTEX range — shard_views(V, rank, world) — so the regulariser
  work is split over the ranks too.
"""
import ctypes
import os
import weakref

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "FMHR_FORCE_DEVICE" in os.environ:            # rehearsal: all ranks on one GPU
        local = int(os.environ["FMHR_FORCE_DEVICE"])
    # FMHR_FORCE_SHARDED=1: rehearsal of the N > 1 code path on ONE rank (RCCL with a world of one on a one-GPU box: the process
    # group, the in-place all-reduce of the step's bucket on the step's stream and the combine kernel are the real ones)
    force = os.environ.get("FMHR_FORCE_SHARDED", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            # FMHR_DIST_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("FMHR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class _NcclUniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_ubyte * 128)]     # (c_char fields read back NUL-terminated: the id would be cut short)


class _NcclConfig(ctypes.Structure):
    """ncclConfig_t up to `splitShare` (the NCCL 2.18 layout; RCCL's parser copies min(size, its own sizeof) bytes over a default-
    initialised struct and resets whatever is newer than `version`, so the shorter, older form is valid for every later library)."""
    _fields_ = [("size", ctypes.c_size_t), ("magic", ctypes.c_uint), ("version", ctypes.c_uint), ("blocking", ctypes.c_int),
                ("cgaClusterSize", ctypes.c_int), ("minCTAs", ctypes.c_int), ("maxCTAs", ctypes.c_int), ("netName", ctypes.c_char_p),
                ("splitShare", ctypes.c_int)]


_NCCL_IN_PROGRESS = 7                                   # ncclInProgress
_UNDEF_INT = -2147483648                                # NCCL_CONFIG_UNDEF_INT


class StreamAllReduce:
    """The step's ONE all-reduce issued straight into RCCL on the STEP'S OWN STREAM (ncclAllReduce, in place, fp32 sum).

    torch.distributed's nccl backend runs every collective on a stream of its own: the call records an event on the current stream,
    makes its stream wait for it, launches, and makes the current stream wait for the result — two cross-stream hand-overs, measured
    at 14.6 us of device time per call with a world of ONE (profiles/r03_small_views.txt), more than a tenth of the 0.12 ms a rank
    of the 8-GPU run spends on its six views.  This class owns a second communicator on the same ranks (its unique id travels over
    the torch group once, at construction) inside the RCCL instance torch itself loaded (torch/lib/librccl.so: one library, two
    communicators), and enqueues the reduction between the backward's last kernel and the combine kernel like any other launch.
    HamStage2.step(group=<this object>) uses it.

    Construction is split into the three phases make_stream_all_reduce negotiates over the torch group, so that NO rank can be left
    waiting inside RCCL for a peer that never arrives:
      prepare()      local only: load the library, bind the entry points, rank 0 draws the unique id — may raise, blocks nobody;
      (the id is broadcast and every rank's "prepared and holds the id" is agreed on with a MIN all-reduce by the caller)
      connect(t)     ncclCommInitRankConfig with blocking = 0: returns at once; ncclCommGetAsyncError is polled until the communicator
                     is up or `t` seconds have passed, then ncclCommAbort — bounded, whatever the peers do.
    StreamAllReduce(group) runs all three for callers that want the object directly (it raises instead of falling back)."""

    def __init__(self, group=None, timeout=30.0, _connect=True):
        if not (dist.is_initialized() and dist.get_backend(group) == "nccl"):
            raise RuntimeError("StreamAllReduce needs an initialised torch.distributed group on the nccl (RCCL) backend")
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.comm, self.lib, self.uid = None, None, None
        if _connect:
            self.prepare()
            dev = torch.device("cuda", torch.cuda.current_device())
            self.set_id(_broadcast_id(self.id_bytes(), group, dev))
            self.connect(timeout)

    # -- phase 1: local
    def prepare(self):
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self.lib = L = ctypes.CDLL(path if os.path.exists(path) else "librccl.so")
        L.ncclGetErrorString.restype = ctypes.c_char_p
        L.ncclGetUniqueId.argtypes = [ctypes.POINTER(_NcclUniqueId)]
        L.ncclGetVersion.argtypes = [ctypes.POINTER(ctypes.c_int)]
        L.ncclCommInitRankConfig.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _NcclUniqueId, ctypes.c_int,
                                             ctypes.POINTER(_NcclConfig)]
        L.ncclCommGetAsyncError.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.ncclCommAbort.argtypes = [ctypes.c_void_p]
        L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_void_p]
        L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        ver = ctypes.c_int(0)
        self._check(L.ncclGetVersion(ctypes.byref(ver)), "ncclGetVersion")
        if ver.value < 21800:
            raise RuntimeError(f"StreamAllReduce: RCCL {ver.value} has no non-blocking communicator initialisation")
        self.uid = _NcclUniqueId()
        if self.rank == 0:
            self._check(L.ncclGetUniqueId(ctypes.byref(self.uid)), "ncclGetUniqueId")

    def id_bytes(self):
        """the 128 bytes of rank 0's unique id (zeros on the other ranks / before prepare())"""
        return ctypes.string_at(ctypes.byref(self.uid), 128) if self.uid is not None else bytes(128)

    def set_id(self, raw):
        ctypes.memmove(ctypes.byref(self.uid), bytes(raw), 128)

    # -- phase 2: the only call that needs the peers — non-blocking, polled against a deadline
    def connect(self, timeout=30.0):
        import time
        L = self.lib
        cfg = _NcclConfig(ctypes.sizeof(_NcclConfig), 0xcafebeef, 21800, 0, _UNDEF_INT, _UNDEF_INT, _UNDEF_INT, None, _UNDEF_INT)
        comm = ctypes.c_void_p()
        rc = L.ncclCommInitRankConfig(ctypes.byref(comm), self.world, self.uid, self.rank, ctypes.byref(cfg))
        if rc not in (0, _NCCL_IN_PROGRESS):
            self._check(rc, "ncclCommInitRankConfig")
        self.comm = comm
        try:
            self._wait(time.monotonic() + float(timeout), "ncclCommInitRankConfig")
        except Exception:
            self.abort()
            raise

    def _wait(self, deadline, what, spin=0.0):
        """poll the communicator's asynchronous state until it leaves ncclInProgress; TimeoutError past the deadline.
        `spin` seconds of busy polling come first (the STEP's all-reduce: on a non-blocking communicator RCCL's helper thread hands the
        enqueue back as ncclInProgress as a matter of course, and a 2 ms sleep there would be ten steps of the 8-GPU run — ADVICE r4);
        after them, and from the start for connect(), the poll sleeps between looks."""
        import time
        state = ctypes.c_int(_NCCL_IN_PROGRESS)
        spin_until = time.monotonic() + spin
        while True:
            self._check(self.lib.ncclCommGetAsyncError(self.comm, ctypes.byref(state)), "ncclCommGetAsyncError")
            if state.value != _NCCL_IN_PROGRESS:
                self._check(state.value, what)
                return
            now = time.monotonic()
            if now > deadline:
                raise TimeoutError(f"StreamAllReduce: {what} still in progress at its deadline")
            if now >= spin_until:
                time.sleep(0.002 if spin == 0.0 else 0.0002)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"StreamAllReduce: {what} failed: {self.lib.ncclGetErrorString(rc).decode()}")

    def all_reduce_(self, t, stream=None):
        """in-place sum of the contiguous float32 GPU tensor `t` over the ranks, enqueued on `stream` (default: the current one)"""
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("StreamAllReduce.all_reduce_: contiguous float32 GPU tensor required")
        st = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
        rc = self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), 7, 0, self.comm, st)   # ncclFloat32, ncclSum
        if rc == _NCCL_IN_PROGRESS:                      # a non-blocking communicator may hand the enqueue back unfinished
            import time
            self._wait(time.monotonic() + 30.0, "ncclAllReduce", spin=0.005)          # busy-poll: the enqueue completes within microseconds
        else:
            self._check(rc, "ncclAllReduce")
        return t

    def gave_up(self):
        """True if the communicator reports an asynchronous error (ncclCommGetAsyncError: a peer died, a transport failed) — what
        ExchangeMonitor polls; ncclInProgress (an enqueue still being handed over) is not an error"""
        if not getattr(self, "comm", None):
            return True
        state = ctypes.c_int(0)
        rc = self.lib.ncclCommGetAsyncError(self.comm, ctypes.byref(state))
        return rc != 0 or state.value not in (0, _NCCL_IN_PROGRESS)

    def abort(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommAbort(self.comm)
            self.comm = None

    def close(self):
        """orderly end of a communicator every rank still holds: ncclCommFinalize (non-blocking communicators must be finalised
        before they are destroyed), polled against a short deadline, then ncclCommDestroy; anything else — abort"""
        if getattr(self, "comm", None):
            import time
            try:
                fin = getattr(self.lib, "ncclCommFinalize", None)
                if fin is not None:
                    fin.argtypes = [ctypes.c_void_p]
                    rc = fin(self.comm)
                    if rc not in (0, _NCCL_IN_PROGRESS):
                        self._check(rc, "ncclCommFinalize")
                    self._wait(time.monotonic() + 5.0, "ncclCommFinalize")
                self.lib.ncclCommDestroy(self.comm)
                self.comm = None
            except Exception:  # noqa: BLE001
                self.abort()


def _sync_bounded(dev, deadline):
    """wait for the current stream of `dev` WITH a deadline (torch.cuda.synchronize() has none): an event polled with query()"""
    import time
    if dev.type != "cuda":
        return
    ev = torch.cuda.Event()
    ev.record()
    while not ev.query():
        if time.monotonic() > deadline:
            raise TimeoutError("the direct all-reduce of the test vector did not complete before the deadline")
        time.sleep(0.001)


def _broadcast_id(raw, group, dev):
    """rank 0's 128 id bytes to every rank of `group` as ONE tensor broadcast (a collective every rank enters, prepared or not)"""
    t = torch.tensor(list(raw), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return bytes(t.cpu().tolist())


def _agree(ok, group, dev):
    """MIN over the ranks of a local yes / no"""
    flag = torch.tensor([1.0 if ok else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return float(flag) == 1.0


def negotiate_stream_all_reduce(group, dev, factory, timeout=30.0, check_numel=4099, log=None):
    """The protocol of make_stream_all_reduce on any backend (`factory(group)` -> an object with prepare / id_bytes / set_id / connect /
    all_reduce_ / abort / close; the CPU test drives it over gloo with fakes that fail on one rank).  Every collective below is entered
    by EVERY rank whatever happened to it locally, and the one call that waits for peers inside RCCL (connect) is only entered once all
    ranks have said they are ready for it, and is bounded by `timeout` on each of them:
      1. prepare() locally; 2. broadcast of the id; 3. MIN(prepared) — any 0: everybody returns the torch group, nobody connects;
      4. connect(timeout) — non-blocking init polled against the deadline, aborted on expiry; 5. MIN(connected);
      6. the direct path must reproduce torch's all-reduce of a test vector; 7. MIN(verified).
    Returns (object or `group`, reason) — reason is None on success, else the first phase that failed somewhere."""
    say = log or (lambda m: None)
    sar, ok = None, True
    try:
        sar = factory(group)
        sar.prepare()
    except Exception as e:  # noqa: BLE001
        ok = False
        say(f"prepare failed on this rank: {e}")
    raw = _broadcast_id(sar.id_bytes() if (ok and sar is not None) else bytes(128), group, dev)
    if ok:
        try:
            sar.set_id(raw)
        except Exception as e:  # noqa: BLE001
            ok = False
            say(f"set_id failed on this rank: {e}")
    if not _agree(ok, group, dev):
        return group, "prepare"
    try:
        sar.connect(timeout)
    except Exception as e:  # noqa: BLE001
        ok = False
        say(f"connect failed on this rank: {e}")
    if not _agree(ok, group, dev):
        if ok:
            sar.abort()                                  # this rank's communicator came up, a peer's did not: give it back
        return group, "connect"
    try:
        g = torch.Generator(device="cpu").manual_seed(1234 + dist.get_rank(group))
        x = torch.randn(check_numel, generator=g).to(dev)
        y = x.clone()
        dist.all_reduce(x, group=group)
        sar.all_reduce_(y)
        # bounded: a rank whose direct all-reduce never completes (a peer failed inside it) must still reach the MIN below — it aborts
        # its communicator there, which also releases the enqueued collective (torch.cuda.synchronize() has no deadline)
        import time
        _sync_bounded(dev, time.monotonic() + float(timeout))
        ok = bool(torch.allclose(x, y, rtol=1e-6, atol=1e-6))
    except Exception as e:  # noqa: BLE001
        ok = False
        say(f"verification failed on this rank: {e}")
    if not ok:
        sar.abort()                                      # BEFORE the agreement: it frees this rank's stream if the collective is stuck
    if not _agree(ok, group, dev):
        sar.abort()
        return group, "verify"
    return sar, None


class PeerAllReduce:
    """The step's ONE exchange as a one-shot peer-to-peer all-reduce over the direct xGMI links (fmhr_amd/csrc/peer.hip:
    fmhr_peer_*): every rank exports a block of uncached device memory by IPC handle, maps the peers' blocks, and one launch per rank on
    the STEP'S OWN STREAM stages the bucket, reduces the rank's shard straight out of the peers' memory and gathers the reduced shards —
    two flag exchanges instead of the 2 (G - 1) dependent hops of a ring; every wait inside the kernel is bounded.  Same calling surface
    as StreamAllReduce (`all_reduce_(tensor, stream)`, `close()`), so HamStage2.step(group=<this object>) uses it as it stands.  The
    tensor must be the `numel` float32 elements the object was negotiated for (the step's bucket: 12 V + 8).
    Set up by negotiate_peer_all_reduce — phase by phase over the torch group, like the RCCL communicator."""

    def __init__(self, group, numel):
        from . import _lib
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.numel = int(numel)
        self._lib = _lib
        self.state = ctypes.c_void_p()
        self.hbytes = int(_lib.load().fmhr_peer_handle_bytes())

    def create(self):
        self._lib.call("fmhr_peer_create", self.rank, self.world, self.numel, ctypes.byref(self.state))

    def handle(self):
        buf = ctypes.create_string_buffer(self.hbytes)
        self._lib.call("fmhr_peer_export", self.state, buf)
        return bytes(buf.raw)

    def import_peers(self, handles):
        for p, h in enumerate(handles):
            if p != self.rank:
                self._lib.call("fmhr_peer_import", self.state, p, ctypes.create_string_buffer(bytes(h), self.hbytes))

    def all_reduce_(self, t, stream=None):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != self.numel:
            raise RuntimeError(f"PeerAllReduce.all_reduce_: the contiguous float32 GPU tensor of {self.numel} elements it was set up for is required")
        st = torch.cuda.current_stream(t.device).cuda_stream if stream is None else stream
        st = st.value if isinstance(st, ctypes.c_void_p) else st
        self._lib.call("fmhr_peer_all_reduce", self.state, ctypes.c_void_p(t.data_ptr()), self.numel, ctypes.c_void_p(st))
        return t

    def staging(self, owner):
        """The staging area of this rank's exported block as a float32 tensor of `numel` elements, for ONE owner (the first to ask; None
        for anyone else): a step that writes its bucket there and passes it to all_reduce_ has nothing to stage — the launch drops its
        copy pass and one of its two grid barriers (fmhr_peer_staging).  The memory is uncached device memory owned by this object: the
        tensor must not outlive it (close() is for the end of the run)."""
        if getattr(self, "_stage_owner", None) is None:
            ptr, n = ctypes.c_void_p(), ctypes.c_longlong(0)
            self._lib.call("fmhr_peer_staging", self.state, ctypes.byref(ptr), ctypes.byref(n))

            class _Block:                                                # (torch maps device memory it does not own through this protocol)
                __cuda_array_interface__ = dict(shape=(int(n.value),), typestr="<f4", data=(int(ptr.value), False), version=2)
            self._stage_keep = _Block()
            self._stage = torch.as_tensor(self._stage_keep, device=torch.device("cuda", torch.cuda.current_device()))
            self._stage_owner = weakref.ref(owner)
        return self._stage if self._stage_owner() is owner else None

    def gave_up(self):
        """True if a wait inside any all-reduce so far ran into its bound (synchronises): the object must not be used again"""
        flag = ctypes.c_int(0)
        self._lib.call("fmhr_peer_status", self.state, ctypes.byref(flag))
        return flag.value != 0

    def abort(self):
        self.close()

    def close(self):
        if self.state:
            self._lib.load().fmhr_peer_destroy(self.state)
            self.state = ctypes.c_void_p()


def negotiate_peer_all_reduce(group, dev, numel, factory=None, timeout=30.0, log=None):
    """PeerAllReduce for `group` or None, agreed phase by phase — every collective below is a torch collective every rank enters whatever
    happened to it locally, and nothing in between waits for a peer on the host:
      1. create() locally (allocate + export);  2. all_gather of the 64-byte handles;  3. MIN(created);
      4. import the peers' blocks (local calls);  5. MIN(imported);
      6. the one-shot all-reduce of a test vector must reproduce torch's — its completion is awaited with a DEADLINE (the kernel's own
         waits are bounded too);  7. MIN(verified).
    Returns (object or None, reason)."""
    say = log or (lambda m: None)
    world = dist.get_world_size(group)
    par, ok = None, True
    try:
        par = (factory or PeerAllReduce)(group, numel)
        par.create()
        mine = par.handle()
    except Exception as e:  # noqa: BLE001
        ok, mine = False, bytes(64)
        say(f"peer all-reduce: create failed on this rank: {e}")
    hb = len(mine)
    hdev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")        # (gloo gathers host tensors only)
    t = torch.tensor(list(mine), dtype=torch.uint8, device=hdev)
    allh = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allh, t, group=group)
    if not _agree(ok, group, dev):
        if par is not None:
            par.close()                                  # whatever this rank got as far as allocating (close() is idempotent) — ADVICE r5
        return None, "peer-create"
    try:
        par.import_peers([bytes(h.cpu().tolist())[:hb] for h in allh])
    except Exception as e:  # noqa: BLE001
        ok = False
        say(f"peer all-reduce: import failed on this rank: {e}")
    if not _agree(ok, group, dev):
        par.close()
        return None, "peer-import"
    try:
        import time
        g = torch.Generator(device="cpu").manual_seed(4321 + dist.get_rank(group))
        x = torch.randn(numel, generator=g).to(dev)
        y = x.clone()
        dist.all_reduce(x, group=group)
        par.all_reduce_(y)
        _sync_bounded(dev, time.monotonic() + float(timeout))
        ok = (not par.gave_up()) and bool(torch.allclose(x, y, rtol=1e-5, atol=1e-5))
        # ... and three more with fresh data each: the flags carry the call count, and a rank that served a peer's block of the
        # PREVIOUS call out of a cache would show here
        for k in range(3):
            x = torch.randn(numel, generator=g).to(dev)
            y = x.clone()
            dist.all_reduce(x, group=group)
            par.all_reduce_(y)
            _sync_bounded(dev, time.monotonic() + float(timeout))
            ok = ok and (not par.gave_up()) and bool(torch.allclose(x, y, rtol=1e-5, atol=1e-5))
    except Exception as e:  # noqa: BLE001
        ok = False
        say(f"peer all-reduce: verification failed on this rank: {e}")
    if not _agree(ok, group, dev):
        par.close()
        return None, "peer-verify"
    return par, None


def make_stream_all_reduce(group=None, check_numel=4099, timeout=None, log=None, numel=None):
    """-> the exchange every rank of `group` could set up for the step's bucket of `numel` floats: a StreamAllReduce (RCCL on the step's
    stream) — the DEFAULT on the nccl backend —, else `group` itself (the torch path); with FMHR_PEER_ALLREDUCE=1 a PeerAllReduce (one-shot
    peer-to-peer over the direct links; needs `numel`) is negotiated first.  The peer-to-peer kernel is OPT-IN since round 6: it has
    only ever run between processes on ONE GPU, where the "remote" blocks sit behind the same L2; whether IPC-mapped uncached memory of
    another device is visible inside a running kernel the way it assumes cannot be rehearsed on a one-GPU box, so the path a first multi-GPU
    run takes by default is the library's.  Whichever path is taken, fmhr_amd.dist.ExchangeMonitor checks it while the loop runs.
    The RCCL communicator cannot be had on EVERY rank:
    negotiate_stream_all_reduce above — agreed phase by phase over the torch group, the RCCL initialisation non-blocking and bounded
    (FMHR_STREAM_ALLREDUCE_TIMEOUT seconds, default 30), no rank ever left inside it.  FMHR_STREAM_ALLREDUCE=0 switches it off.
    `make_stream_all_reduce.last` = (path, reason) of the last call, for the bench line."""
    make_stream_all_reduce.last = ("torch", "disabled or not the nccl backend")
    if os.environ.get("FMHR_STREAM_ALLREDUCE", "1") == "0" or not dist.is_initialized():
        return group
    if timeout is None:
        timeout = float(os.environ.get("FMHR_STREAM_ALLREDUCE_TIMEOUT", "30"))
    if dist.get_backend(group) != "nccl":
        # a rehearsal on one GPU (ranks over gloo, FMHR_FORCE_DEVICE): the peer-to-peer exchange needs no RCCL — hipIpc handles work between
        # processes on one device — so it can be rehearsed there when asked for EXPLICITLY (FMHR_PEER_ALLREDUCE=1); the fallback is the group
        if numel is not None and os.environ.get("FMHR_PEER_ALLREDUCE") == "1" and torch.cuda.is_available():
            par, peer_why = negotiate_peer_all_reduce(group, torch.device("cuda", torch.cuda.current_device()), int(numel), timeout=timeout, log=log)
            if par is not None:
                make_stream_all_reduce.last = ("peer-to-peer one-shot on the step's stream", "rehearsal: ranks over " + dist.get_backend(group))
                return par
            make_stream_all_reduce.last = ("torch", f"not the nccl backend (peer-to-peer: {peer_why})")
        return group
    dev = torch.device("cuda", torch.cuda.current_device())
    peer_why = "opt-in (FMHR_PEER_ALLREDUCE=1): never run across two devices"
    if numel is not None and os.environ.get("FMHR_PEER_ALLREDUCE", "0") == "1":
        par, peer_why = negotiate_peer_all_reduce(group, dev, int(numel), timeout=timeout, log=log)
        if par is not None:
            make_stream_all_reduce.last = ("peer-to-peer one-shot on the step's stream", None)
            return par
    out, why = negotiate_stream_all_reduce(group, dev, lambda g: StreamAllReduce(g, _connect=False), timeout, check_numel, log)
    make_stream_all_reduce.last = (("rccl-on-step-stream", f"peer-to-peer not used: {peer_why}") if why is None else
                                   ("torch", f"fell back at phase '{why}' (peer-to-peer: {peer_why})"))
    return out


make_stream_all_reduce.last = ("torch", "not called")


def shard_views(n_views, rank, world):
    """Contiguous, balanced partition: returns (start, stop) of this rank's views."""
    base, rem = divmod(n_views, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(n_batch, V, rank, world):
    """ONE minibatch of the loop (`perm[k:k+batch]`, mesh_sfs_optim.py:255-260: n_batch views, the last batch of an epoch may be ragged)
    over the ranks: the rank's contiguous share of the batch's views, and its share of the view-independent terms.  A batch with fewer
    views than ranks leaves ranks WITHOUT a view: they render nothing and enter the step's exchange with a zero share (the exchange is a
    collective: every rank takes part in every step); the regularisers are split by vertex range over the ranks that do have views.
    Returns dict(views=(s0, s1) within the batch, n_local, active = ranks with views, reg_range=(v0, v1))."""
    s0, s1 = shard_views(n_batch, rank, world)
    active = min(world, n_batch)
    return dict(views=(s0, s1), n_local=s1 - s0, active=active, reg_range=shard_views(V, rank, active) if rank < active else (0, 0))


class ExchangeError(RuntimeError):
    """the ranks no longer hold the same parameters, or the step's exchange reported a failure"""


class ExchangeMonitor:
    """What makes a dead exchange impossible to miss.  The replicated parameters of a view-sharded run stay
    identical ONLY because every rank applies the same update to the same all-reduced gradient; an exchange that silently returns a rank's
    own share (a peer-to-peer wait that ran into its bound, a communicator in an error state) would let the ranks optimise apart — and read
    as a FASTER run.  `verify(tensors)` judges the EXCHANGE, and reports on the PARAMETERS:
      1. the exchange's own failure state — PeerAllReduce.gave_up(): the kernel's sticky abort word; StreamAllReduce.gave_up(): RCCL's
         asynchronous error;
      2. a PROBE through the step's own exchange path: a test vector that differs on every rank is reduced by the exchange object AND by
         torch's all-reduce on the torch group; they must agree (an exchange that hands back the rank's own data does not);
      3. a checksum of the BITS of the replicated tensors (sum of their int32 images, in int64), MAX and MIN over the ranks through the torch
         group, and a NaN flag; only when the bits differ: the spread over the ranks (all-reduce MAX / MIN of the tensors themselves) in the
         L2 norm and in the max norm, relative to the tensors' own.
    Bits that differ are not by themselves an error: an all-reduce need not be bit-reproducible ACROSS ranks, and behind an L1 loss a 1e-7
    difference flips the sign of the odd residual, which Adam turns into O(lr) per step for that vertex (measured: DESIGN_LOG.md §9) — that
    is reported (`same_bits`, `spread_l2`, `spread_max`) and, with `resync=[...]`, repaired by broadcasting rank 0's tensors.  ExchangeError is
    raised ON EVERY RANK (the verdict is formed from all-reduced quantities) when an exchange failed, the probe disagrees, a value is not
    finite, or the ranks are apart by more than `tol` in L2 (default 0.25: ranks that stopped exchanging optimise different problems).
    A few small collectives + one synchronisation: the loops call it every `check_every` epochs and at their end, bench.py after its timed
    region."""

    def __init__(self, group, exchange=None, dev=None, tol=0.25):
        self.group, self.exchange, self.tol = group, exchange, float(tol)
        self.dev = dev if dev is not None else torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.checks, self.last, self.resyncs = 0, None, 0

    def _active(self):
        return self.group is not None and dist.is_initialized() and dist.get_world_size(self.group) > 1

    @staticmethod
    def fingerprint(tensors):
        return torch.stack([t.detach().contiguous().view(-1).view(torch.int32).to(torch.int64).sum() for t in tensors])

    def probe(self):
        """1 if the step's exchange reduces a rank-dependent test vector exactly as torch's all-reduce does (or there is no exchange object
        of its own), else 0 — local verdict; verify() all-reduces it"""
        ex = self.exchange
        if ex is None or not hasattr(ex, "all_reduce_") or not self._active():
            return 1
        try:
            rank = dist.get_rank(self.group)
            numel = int(getattr(ex, "numel", 4096))
            x = ((torch.arange(numel, device=self.dev, dtype=torch.float32) % 251.0) + 1.0) * float(rank + 1) * 2.0 ** -8   # (exact in fp32, sums exact)
            y = x.clone()
            dist.all_reduce(x, group=self.group)
            ex.all_reduce_(y)
            if self.dev.type == "cuda":
                torch.cuda.synchronize(self.dev)
            return 1 if bool(torch.equal(x, y)) else 0
        except Exception:  # noqa: BLE001
            return 0

    def verify(self, tensors, what="parameters", resync=None):
        """-> dict(ok, same_bits, spread_l2, spread_max, exchange_failed, probe_ok, non_finite, checks, resyncs); raises ExchangeError on every rank when not ok"""
        bad_local = 0
        ex = self.exchange
        if ex is not None and hasattr(ex, "gave_up"):
            try:
                bad_local = 1 if ex.gave_up() else 0
            except Exception:  # noqa: BLE001
                bad_local = 1
        probe_bad = 0 if self.probe() else 1                            # (entered by EVERY rank, whatever its own state: the probe holds a collective)
        k = len(tensors)
        fp = self.fingerprint(tensors).to(self.dev)
        nan = torch.stack([(~torch.isfinite(t.detach())).any().to(torch.int64) for t in tensors]).to(self.dev)
        v = torch.cat([fp, -fp, nan, torch.tensor([bad_local, probe_bad], dtype=torch.int64, device=self.dev)])
        if self._active():
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self.group)
        v = v.tolist()
        same = all(v[i] == -v[k + i] for i in range(k))
        non_finite, failed, probe_ok = bool(any(v[2 * k: 3 * k])), bool(v[-2]), not bool(v[-1])
        s_l2 = s_max = 0.0
        if not same and not non_finite and self._active():           # (every rank knows `same`: all of them enter these collectives)
            for t in tensors:
                hi, lo = t.detach().clone(), t.detach().clone()
                dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
                dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
                mag = torch.maximum(hi.abs(), lo.abs()).double()
                d = (hi - lo).double()
                s_l2 = max(s_l2, float(d.norm() / mag.norm().clamp_min(1e-300)))
                s_max = max(s_max, float(d.abs().max() / mag.max().clamp_min(1e-300)))
        self.checks += 1
        ok = (not failed) and probe_ok and (not non_finite) and (same or s_l2 <= self.tol)
        self.last = dict(ok=ok, same_bits=same, spread_l2=s_l2, spread_max=s_max, exchange_failed=failed, probe_ok=probe_ok, non_finite=non_finite,
                         checks=self.checks, resyncs=self.resyncs)
        if not ok:
            rank = dist.get_rank(self.group) if self._active() else 0
            raise ExchangeError(f"rank {rank}: the view-sharded run is broken ({what}): "
                                + ("the step's exchange reported a failure on some rank (a bounded wait gave up / RCCL asynchronous error); " if failed else "")
                                + ("the step's exchange does not reduce a test vector the way torch's all-reduce does; " if not probe_ok else "")
                                + (f"the ranks hold DIFFERENT values (spread {s_l2:.2e} in L2, {s_max:.2e} in the max norm); " if not same and s_l2 > self.tol else "")
                                + ("non-finite values; " if non_finite else "")
                                + f"check {self.checks} — results since the previous check are invalid")
        if not same and resync and self._active():                   # close but not the same bits: rank 0's values for everybody
            src = dist.get_global_rank(self.group, 0) if self.group is not dist.group.WORLD else 0
            for t in resync:
                dist.broadcast(t.detach(), src=src, group=self.group)
            self.resyncs += 1
            self.last["resyncs"] = self.resyncs
        return self.last


def shard_frames_views(n_frames, n_views, rank, world):
    """Partition of a SEQUENCE capture (BASELINE.json configs[4]: 8 frames x 32 views, conf/ih_sfsseq.conf) over `world` ranks,
    frames first, then views.  Every frame is an independent (delta, albedo) problem on the shared topology, so ranks that hold
    different frames never exchange anything; only when there are more ranks than frames do the ranks of one frame split its
    views and all-reduce inside the frame's own subgroup (the single-collective step of HamStage2.step(group=...)).

    Returns dict(frames=(f0, f1), views=(v0, v1), peers=[ranks sharing this rank's frame] or None, view_rank=i, view_world=k):
      world <= n_frames : frames split contiguously (balanced), every rank renders all views of its frames, peers = None;
      world  > n_frames : world must be a multiple of n_frames; rank r works on frame r // k (k = world / n_frames) and on the
                          view share shard_views(n_views, r % k, k); peers = the k ranks of that frame."""
    if world <= n_frames:
        f0, f1 = shard_views(n_frames, rank, world)
        return dict(frames=(f0, f1), views=(0, n_views), peers=None, view_rank=0, view_world=1)
    if world % n_frames != 0:
        raise ValueError(f"shard_frames_views: {world} ranks cannot be split evenly over {n_frames} frames")
    k = world // n_frames
    f = rank // k
    v0, v1 = shard_views(n_views, rank % k, k)
    return dict(frames=(f, f + 1), views=(v0, v1), peers=list(range(f * k, (f + 1) * k)), view_rank=rank % k, view_world=k)


def frame_groups(n_frames, world):
    """The process groups of shard_frames_views for world > n_frames: every rank must create ALL of them, in the same order
    (torch.distributed.new_group is collective).  Returns a list indexed by frame (None when no exchange is needed)."""
    if world <= n_frames:
        return [None] * n_frames
    k = world // n_frames
    return [dist.new_group(list(range(f * k, (f + 1) * k))) for f in range(n_frames)]


class ShardedStage1:
    """Stage 1 (mesh_sfs_optim.py:196-240: albedo + SH, vertices frozen) with the views sharded like stage 2: every rank builds the
    HamStage1 plan of ITS views and keeps their SH rows; an iteration is the rank's fused step, one all-reduce(SUM) of the albedo
    gradient [V,3] (0.59 MB at V = 49 k) and the replicated Adam update.  F.l1_loss divides by 3 x the valid pixels of the WHOLE
    batch (:231): the valid-pixel count of every view is known when the plans are built (one all-gather, once), so each rank
    rescales its share by Pv_local / Pv_global BEFORE the exchange — no second collective."""

    def __init__(self, hs1, view0, n_views_global, group=None):
        """group: a torch.distributed group, or a StreamAllReduce on one (the iteration's all-reduce then runs on the step's stream)"""
        self.sar = group if hasattr(group, "all_reduce_") else None
        if self.sar is not None:
            group = self.sar.group if self.sar.group is not None else dist.group.WORLD
        self.hs1, self.group, self.view0, self.n_global = hs1, group, int(view0), int(n_views_global)
        counts = torch.zeros(self.n_global, dtype=torch.float32, device=hs1.dev)
        counts[self.view0: self.view0 + hs1.n] = hs1.view_valid.to(torch.float32)
        if group is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(counts, group=group)
        self.counts = counts                                      # valid pixels of every view of the capture

    def step_sel(self, albedo, sh_local, sel_global):
        """The same iteration with the batch given as a 0 / 1 mask over the capture's views (int32 [n_views_global] on the GPU — a row of
        fmhr_amd.pipeline.PermChunks' masks): no index kernels, no host synchronisation; a rank none of whose views is in the batch runs
        the step on an all-zero selection (zero gradient, zero loss share) and enters the all-reduce like everybody else.
        Returns (loss, galbedo [V,3] summed over the ranks, gsh [n_local,9])."""
        hs1 = self.hs1
        local = sel_global[self.view0: self.view0 + hs1.n]
        w = sel_global.to(torch.float32)
        pv_g = (self.counts * w).sum()
        pv_l = (hs1.view_valid.to(torch.float32) * w[self.view0: self.view0 + hs1.n]).sum()
        hs1.step(albedo, sh_local, sel=local)
        return self._exchange(pv_l / pv_g.clamp_min(1.0))

    def _exchange(self, scale):
        hs1 = self.hs1
        galbedo = hs1.galbedo * scale
        gsh = hs1.gsh * scale
        loss = (hs1.loss_view.sum() * scale).reshape(1)
        if self.group is not None and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            buf = torch.cat([galbedo.reshape(-1), loss])
            if self.sar is not None:
                self.sar.all_reduce_(buf)                         # the iteration's ONE collective, on the current stream
            else:
                dist.all_reduce(buf, group=self.group)
            galbedo, loss = buf[:-1].view_as(hs1.galbedo), buf[-1:]
        return loss[0], galbedo, gsh

    def step(self, albedo, sh_local, idx_global=None):
        """idx_global: the batch as GLOBAL view indices (None: all views).  Returns (loss, galbedo [V,3] summed over the ranks,
        gsh [n_local,9] for this rank's SH rows); the caller applies the optimiser (the same update on every rank)."""
        hs1 = self.hs1
        if idx_global is None:
            local, pv_g = None, self.counts.sum()
        else:
            ig = idx_global.to(hs1.dev).long()
            pv_g = self.counts[ig].sum()
            mine = ig[(ig >= self.view0) & (ig < self.view0 + hs1.n)] - self.view0
            local = mine
        if local is not None and local.numel() == 0:
            hs1.galbedo.zero_(); hs1.gsh.zero_(); hs1.loss_view.zero_()
            scale = torch.zeros((), device=hs1.dev)
        else:
            hs1.step(albedo, sh_local, idx=local)
            pv_l = hs1.view_valid.to(torch.float32).sum() if local is None else hs1.view_valid[local].to(torch.float32).sum()
            scale = pv_l / pv_g.clamp_min(1.0)
        return self._exchange(scale)


def combine_bucket(bucket, V):
    """The arithmetic of fmhr_ham_combine on any device (the GPU step uses the kernel; this is the host-side statement the CPU
    rehearsal checks): bucket = [P_delta | P_albedo | R_delta | R_albedo | 8 loss slots] summed over the ranks."""
    n3 = 3 * V
    losses = bucket[4 * n3:]
    nv = float(losses[7])
    sc = 1.0 / (3.0 * nv) if nv > 0 else 0.0
    g = bucket[: 2 * n3] * sc + bucket[2 * n3: 4 * n3]
    losses = losses.clone()
    losses[0] = losses[0] * sc
    losses[6] = losses[0] + losses[1] + losses[2] + losses[3] + losses[4] + losses[5]
    return losses, g[:n3].view(V, 3), g[n3:].view(V, 3)


class ShardedStep:
    """The single-collective exchange on plain tensors (any device): pack the rank's shares, all-reduce once, combine.
    HamStage2.step(group=...) does the same in place on its own bucket with the combine kernel; this class is the host-side
    reference of the protocol (tests/test_dist_gloo.py runs it over gloo with the CPU oracle as the compute)."""

    def __init__(self, V, device, group=None, exchange=True):
        """exchange=False: this rank holds a whole frame of a sequence capture (shard_frames_views with peers = None): nothing to
        reduce, whatever the size of the default group."""
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_initialized() and exchange) else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bucket = torch.zeros(12 * V + 8, dtype=torch.float32, device=device)
        self.V = V

    def reduce(self, losses, p_delta, p_albedo, r_delta, r_albedo):
        """losses[8]: this rank's share of every slot — slot 0 = w_sfs * sum |d| over its valid pixels (no divisor), slot 7 = its LOCAL
        valid-pixel count; p_* = photometric gradient share without the divisor, r_* = everything else.  Returns the batch-global
        (losses, gdelta, galbedo)."""
        V, b = self.V, self.bucket
        n3 = 3 * V
        for k, t in enumerate((p_delta, p_albedo, r_delta, r_albedo)):
            b[k * n3: (k + 1) * n3].copy_(t.reshape(-1))
        b[4 * n3:].copy_(losses)
        if self.world > 1:
            dist.all_reduce(b, group=self.group)
        return combine_bucket(b, V)
