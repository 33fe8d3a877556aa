"""Multi-GPU sharding of the loop-closure hot path.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in
the CPU tests).  The reference has no collective at all; what shards here:
  * descriptor generation : scans are split contiguously over ranks, no communication;
  * descriptor database   : every rank contributes the descriptors it just built with ONE
                            all-gather (ragged shards are padded to the longest), after which
                            every rank holds the whole database, exactly like the per-robot
                            Python lists of RING_ros/main_RING.py:285-289 but device resident;
  * candidate scoring     : the (query, candidate) pair list / the GICP pair list is split
                            contiguously; only the tiny result rows are gathered;
  * alternatives that keep the database sharded: sharded_topk_sweep (queries travel, top-k rows
                            come back) and fetch_rows (only the candidate rows asked for travel).
No kernel contains an exchange step.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced split: the first n_items % world ranks get one extra item."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world):
    return [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]


def allgather_ragged(local, group=None):
    """All-gather of per-rank tensors whose leading dimension differs.  Returns the
    concatenation in rank order (every rank gets the same tensor)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    if longest == 0:
        return local
    padded = local
    if local.shape[0] < longest:
        pad = torch.zeros((longest - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    out = torch.empty((world * longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == longest for c in counts):
        return out
    return torch.cat([out[r * longest: r * longest + counts[r]] for r in range(world)], 0)


def sharded_sweep(queries, local_db, sweep_fn, group=None):
    """Score `queries` (same on every rank) against the database whose rows are spread over the
    ranks.  Alternative to gathering the database: only the [Q, n_local] result rows travel.
    sweep_fn(queries, db) -> (dist [Q,n], angle [Q,n]).  Returns full-width (dist, angle)."""
    d, a = sweep_fn(queries, local_db) if local_db.shape[0] else (
        torch.zeros((queries.shape[0], 0), dtype=torch.float32, device=queries.device),
        torch.zeros((queries.shape[0], 0), dtype=torch.int32, device=queries.device))
    d_all = allgather_ragged(d.t().contiguous(), group).t().contiguous()
    a_all = allgather_ragged(a.t().contiguous(), group).t().contiguous()
    return d_all, a_all


def _world(group=None):
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


def _rank(group=None):
    return dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0


def sharded_topk_sweep(queries, local_db, sweep_fn, k, group=None, shard_rows=None, packed=False, exchange=None):
    """The low-traffic alternative to replicating the database: the database stays sharded, the
    queries are the same on every rank (all-gather them first if they are not), every rank scores them against ITS rows
    and only the k best (dist, angle, global row) per query travel.  Rows are numbered in rank order (rank r owns
    [sum of the earlier shards, ...)).  Returns (dist [Q,k], angle [Q,k], row [Q,k]) sorted by ascending dist, identical
    on every rank; ties resolve to the smaller global row, like a single-rank sweep followed by a stable sort.
    shard_rows (rows per rank, the same list on every rank) skips the exchange of the shard sizes and its host
    synchronisation: every shape is then static and the call only enqueues work (timed loops).  packed=True sends the three
    result arrays as ONE collective (an int64 view of (dist bits, angle) + the row); exchange (an Exchange; needs packed and shard_rows): that
    collective goes through the C ABI (mrs_exchange_allgather) on the current stream instead of torch.distributed."""
    world, rank = _world(group), _rank(group)
    Q = queries.shape[0]
    dev = queries.device
    n_local = torch.tensor([local_db.shape[0]], dtype=torch.int64, device=dev)
    if shard_rows is not None:
        counts = [int(c) for c in shard_rows]
        assert len(counts) == world and counts[rank] == local_db.shape[0]
    elif world > 1:
        counts = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(counts, n_local, group=group)
        counts = [int(c.item()) for c in counts]
    else:
        counts = [int(n_local.item())]
    base = sum(counts[:rank])
    if local_db.shape[0]:
        d, a = sweep_fn(queries, local_db)
    else:
        d = torch.zeros((Q, 0), dtype=torch.float32, device=dev); a = torch.zeros((Q, 0), dtype=torch.int32, device=dev)
    kk = min(k, d.shape[1])
    row = torch.arange(d.shape[1], device=dev, dtype=torch.int64)[None].expand(Q, -1) + base
    if kk:
        order = torch.argsort(d, dim=1, stable=True)[:, :kk]
        d, a, row = torch.gather(d, 1, order), torch.gather(a, 1, order), torch.gather(row, 1, order)
    # fixed-size exchange: k slots per rank, unused slots carry +inf
    pd = torch.full((Q, k), float("inf"), dtype=torch.float32, device=dev); pd[:, :kk] = d[:, :kk]
    pa = torch.zeros((Q, k), dtype=torch.int32, device=dev); pa[:, :kk] = a[:, :kk]
    pr = torch.full((Q, k), -1, dtype=torch.int64, device=dev); pr[:, :kk] = row[:, :kk]
    if world > 1 and packed:
        buf = torch.empty((Q, k, 2), dtype=torch.int64, device=dev)          # per slot: int32 angle | float32 dist | int64 row
        b32 = buf.view(torch.int32)                                          # [Q, k, 4]
        b32[..., 0] = pa
        b32[..., 1] = pd.contiguous().view(torch.int32)
        buf[..., 1] = pr
        gb = torch.empty((world * Q, k, 2), dtype=torch.int64, device=dev)
        if exchange is not None:
            exchange.allgather_into(gb, buf)
        else:
            dist.all_gather_into_tensor(gb, buf, group=group)
        gb = gb.view(world, Q, k, 2).permute(1, 0, 2, 3).reshape(Q, world * k, 2).contiguous()
        g32 = gb.view(torch.int32)                                           # [Q, world * k, 4]
        pa = g32[..., 0].contiguous()
        pd = g32[..., 1].contiguous().view(torch.float32)
        pr = gb[..., 1].contiguous()
    elif world > 1:
        gd = torch.empty((world * Q, k), dtype=torch.float32, device=dev)    # concatenation along dim 0 (gloo and RCCL alike)
        ga = torch.empty((world * Q, k), dtype=torch.int32, device=dev)
        gr = torch.empty((world * Q, k), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(gd, pd.contiguous(), group=group)
        dist.all_gather_into_tensor(ga, pa.contiguous(), group=group)
        dist.all_gather_into_tensor(gr, pr.contiguous(), group=group)
        pd = gd.view(world, Q, k).permute(1, 0, 2).reshape(Q, world * k)
        pa = ga.view(world, Q, k).permute(1, 0, 2).reshape(Q, world * k)
        pr = gr.view(world, Q, k).permute(1, 0, 2).reshape(Q, world * k)
    # rank-major concatenation = ascending global row among equal distances -> the stable sort keeps that order
    order = torch.argsort(pd, dim=1, stable=True)[:, :k]
    return torch.gather(pd, 1, order), torch.gather(pa, 1, order), torch.gather(pr, 1, order)


def fetch_rows(local_db, global_rows, shard_rows, group=None):
    """Request-based alternative to replicating the database for candidate scoring: every rank names the GLOBAL rows it needs
    (`global_rows` int64 [M], any order, repeats allowed; rank r owns rows [sum(shard_rows[:r]), sum(shard_rows[:r + 1])) ) and
    receives exactly those rows of the owners' exact entries, in request order.  Moves M rows per rank instead of the whole
    database: with B candidates per launch out of N * B rows that is 1 / N of the all-gather's bytes, at the price of one request
    round trip (three all-to-all collectives: counts, indices, rows).  `shard_rows`: rows per rank, the same list on every rank."""
    world, rank = _world(group), _rank(group)
    dev = local_db.device
    rows = global_rows.to(torch.int64).reshape(-1)
    bounds = torch.tensor([0] + list(shard_rows), dtype=torch.int64, device=dev).cumsum(0)      # [world + 1]
    assert int(bounds[-1]) > 0 and len(shard_rows) == world and local_db.shape[0] == int(shard_rows[rank])
    if rows.numel():
        assert int(rows.min()) >= 0 and int(rows.max()) < int(bounds[-1]), "row outside the database"
    if world == 1:
        return local_db[rows]
    owner = torch.bucketize(rows, bounds[1:], right=True)                       # rank that owns each requested row
    order = torch.argsort(owner, stable=True)                                   # requests grouped by owner, original order within
    send_idx = rows[order].contiguous()
    send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts, group=group)               # how many rows each peer wants from me
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    want = torch.empty(int(sum(rc)), dtype=torch.int64, device=dev)
    dist.all_to_all_single(want, send_idx, output_split_sizes=rc, input_split_sizes=sc, group=group)
    payload = local_db[want - bounds[rank]].contiguous()                        # the rows my peers asked for, in their order
    real = torch.view_as_real(payload) if payload.is_complex() else payload
    got = torch.empty((int(sum(sc)),) + tuple(real.shape[1:]), dtype=real.dtype, device=dev)
    dist.all_to_all_single(got, real.contiguous(), output_split_sizes=sc, input_split_sizes=rc, group=group)
    if payload.is_complex():
        got = torch.view_as_complex(got)
    out = torch.empty_like(got)
    out[order] = got                                                            # back to request order
    return out


class RowFetchPlan:
    """fetch_rows with the request phase done ahead of time.  Candidate rows are usually known long before the entries are needed
    (they come out of a coarse search), so the two request collectives (counts, row indices) and their host synchronisation run
    once, here; fetch() is then ONE all-to-all of exactly the requested rows with static split sizes and no host synchronisation,
    which can be issued launches ahead (async_op=True) and waited for on the consuming stream.
    rank r owns global rows [sum(shard_rows[:r]), sum(shard_rows[:r + 1]))."""

    def __init__(self, global_rows, shard_rows, group=None):
        self.group = group
        world, rank = _world(group), _rank(group)
        dev = global_rows.device
        rows = global_rows.to(torch.int64).reshape(-1)
        bounds = torch.tensor([0] + list(shard_rows), dtype=torch.int64, device=dev).cumsum(0)
        assert len(shard_rows) == world and int(bounds[-1]) > 0
        if rows.numel():
            assert int(rows.min()) >= 0 and int(rows.max()) < int(bounds[-1]), "row outside the database"
        self.n = int(rows.numel())
        self.base = int(bounds[rank])
        self.local_rows = int(shard_rows[rank])
        if world == 1:
            self.want, self.order, self.sc, self.rc = rows, None, [self.n], [self.n]
            return
        owner = torch.bucketize(rows, bounds[1:], right=True)
        self.order = torch.argsort(owner, stable=True)                    # requests grouped by owner, original order within
        send_idx = rows[self.order].contiguous()
        send_counts = torch.bincount(owner, minlength=world).to(torch.int64)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=group)
        self.sc, self.rc = send_counts.tolist(), recv_counts.tolist()
        want = torch.empty(int(sum(self.rc)), dtype=torch.int64, device=dev)
        dist.all_to_all_single(want, send_idx, output_split_sizes=self.rc, input_split_sizes=self.sc, group=group)
        self.want = want - self.base                                       # local indices of the rows my peers asked for, in their order
        self.inv = torch.empty_like(self.order)
        self.inv[self.order] = torch.arange(self.n, device=dev)            # request i sits at position inv[i] of the received block

    def bytes_in(self, row_bytes, rank_local_too=False):
        """payload bytes this rank receives per fetch (rows it owns itself do not cross a link)"""
        rank = _rank(self.group)
        return row_bytes * (self.n if rank_local_too else self.n - int(self.sc[rank] if len(self.sc) > 1 else self.n))

    def fetch(self, local_db, async_op=False):
        """-> (rows in request order) or, with async_op, (work, finish) where finish() returns them once work.wait() was called."""
        assert local_db.shape[0] == self.local_rows
        if self.order is None:                                             # single rank: a local gather
            out = local_db[self.want]
            return (None, lambda: out) if async_op else out
        payload = local_db[self.want].contiguous()
        cplx = payload.is_complex()
        real = torch.view_as_real(payload) if cplx else payload
        got = torch.empty((self.n,) + tuple(real.shape[1:]), dtype=real.dtype, device=real.device)
        work = dist.all_to_all_single(got, real.contiguous(), output_split_sizes=self.sc, input_split_sizes=self.rc, group=self.group,
                                      async_op=async_op)

        def finish():
            g = torch.view_as_complex(got) if cplx else got
            return g[self.inv]
        if async_op:
            self._keep = (payload, real)                                   # alive until the collective has run
            return work, finish
        return finish()


class OwnerRescorer:
    """Exact re-scoring of loop candidates whose replica score is too close to the acceptance threshold to trust.

    Non-owner ranks hold fp16 replicas of a descriptor (the exchange format); the owner keeps the exact fp32 entry.  A
    replica score differs from the exact one by < 2e-3 (tests/test_ring_gpu.py::test_fp16_replica_format), while the
    reference accepts a loop on a hard threshold (RING_ros/main_RING.py:137 `dist < cfg.dist_threshold`, config.py:17).
    So every (query, candidate) whose replica distance lies within `margin` of the threshold is sent to the candidate's
    owner, scored there against the exact entry, and the decision is taken on that number.

    Collectives have static shapes (`slots` requests per rank and round, padded), so every rank issues the same
    sequence whatever its data: per round one all-gather of the requests (row index + the query's fp32 descriptor), one
    all-reduce of the answers and one of the number of candidates still waiting; rounds repeat until no rank has any left, so
    no ambiguous candidate is ever decided on its replica score."""

    def __init__(self, threshold, margin=2e-3, slots=64, group=None):
        self.threshold, self.margin, self.slots, self.group = float(threshold), float(margin), int(slots), group
        # still_ambiguous: entries of the last call's OUTPUT whose replica score was ambiguous and that were NOT replaced by an exact one (measured;
        # the loop below only ends when an all-reduce(MAX) of the per-rank remaining counts is 0)
        self.stats = {"calls": 0, "requested": 0, "rounds": 0, "flipped": 0, "still_ambiguous": None}

    def ambiguous(self, dist_replica):
        return torch.nonzero((dist_replica - self.threshold).abs() < self.margin).reshape(-1)

    def rescore(self, dist_replica, angle_replica, cand_row, query_desc, owner_of_row, local_row_of, exact_pair_fn):
        """dist_replica/angle_replica/cand_row: [P] results against replicas and the candidates' GLOBAL rows;
        query_desc: [P, ...] the queries' exact descriptors (device); owner_of_row(rows)->rank tensor and
        local_row_of(rows)->owner-local index tensor map global rows; exact_pair_fn(query_desc [m,...], local_rows [m]) ->
        (dist [m], angle [m]) scores against the caller's OWN exact entries.  Returns (dist, angle) with the ambiguous
        entries replaced by the owners' exact values."""
        world, rank = _world(self.group), _rank(self.group)
        dev = dist_replica.device
        S = self.slots
        amb = self.ambiguous(dist_replica)
        self.stats["calls"] += 1
        self.stats["requested"] += int(amb.numel())
        if amb.numel() > S:                                  # the closest to the threshold first
            amb = amb[torch.argsort((dist_replica[amb] - self.threshold).abs())]
        out_d, out_a = dist_replica.clone(), angle_replica.clone()
        replaced = torch.zeros(dist_replica.shape[0], dtype=torch.bool, device=dev)
        all_amb = amb.clone()
        # fixed-size rounds of `slots` requests per rank until NO rank has an ambiguous candidate left: every rank issues the
        # same sequence of collectives (the number of rounds is agreed on by an all-reduce of the remaining counts)
        while True:
            take, amb = amb[:S], amb[S:]
            m = int(take.numel())
            rows = torch.full((S,), -1, dtype=torch.int64, device=dev)
            rows[:m] = cand_row[take].to(torch.int64)
            q = torch.zeros((S,) + tuple(query_desc.shape[1:]), dtype=query_desc.dtype, device=dev)
            q[:m] = query_desc[take]
            if world > 1:
                all_rows = torch.empty((world * S,), dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(all_rows, rows, group=self.group)
                qr = torch.view_as_real(q) if q.is_complex() else q
                all_q = torch.empty((world * S,) + tuple(qr.shape[1:]), dtype=qr.dtype, device=dev)
                dist.all_gather_into_tensor(all_q, qr.contiguous(), group=self.group)
                if q.is_complex():
                    all_q = torch.view_as_complex(all_q)
            else:
                all_rows, all_q = rows, q
            ans_d = torch.zeros((world * S,), dtype=torch.float32, device=dev)
            ans_a = torch.zeros((world * S,), dtype=torch.int32, device=dev)
            valid = all_rows >= 0
            mine = torch.nonzero(valid & (owner_of_row(all_rows.clamp(min=0)) == rank)).reshape(-1)
            if mine.numel():
                d, a = exact_pair_fn(all_q[mine], local_row_of(all_rows[mine]))
                ans_d[mine] = d.to(torch.float32); ans_a[mine] = a.to(torch.int32)
            left = torch.tensor([int(amb.numel())], dtype=torch.int64, device=dev)
            if world > 1:                                        # exactly one rank fills each slot
                dist.all_reduce(ans_d, group=self.group)
                dist.all_reduce(ans_a, group=self.group)
                dist.all_reduce(left, op=dist.ReduceOp.MAX, group=self.group)
            if m:
                new_d = ans_d[rank * S: rank * S + m]
                self.stats["flipped"] += int(((new_d < self.threshold) != (dist_replica[take] < self.threshold)).sum())
                out_d[take] = new_d
                out_a[take] = ans_a[rank * S: rank * S + m]
                replaced[take] = True
            self.stats["rounds"] = self.stats.get("rounds", 0) + 1
            if int(left.item()) == 0:
                break
        self.stats["still_ambiguous"] = int((~replaced[all_amb]).sum()) if all_amb.numel() else 0
        return out_d, out_a


class Exchange:
    """The exchange step behind the C ABI (include/mrslam_hip.h: mrs_exchange_*, RCCL looked up at run time): what a C++ host would call,
    usable from here as well.  The communicator is created by the library from a unique id that rank 0 makes and torch.distributed (any
    backend) hands to the others; without an initialised process group the world is this one process.
      allgather(local)              -> [world * n, ...]: every rank's `local` ([n, ...], the same n everywhere), in rank order
      fetch_rows(local_db, rows)    -> local_db-shaped rows by GLOBAL index (rank r owns rows [r * n, (r + 1) * n)), in request order"""

    def __init__(self, device=0, group=None):
        import ctypes as C
        from . import _lib
        self._C, self._lib_mod = C, _lib
        lib = _lib.load()
        if not lib.mrs_exchange_available():
            raise _lib.MrsError("RCCL entry points not found (librccl.so.1): the C-ABI exchange is unavailable")
        self.device = int(device)
        self.world, self.rank = _world(group), _rank(group)
        ident = (C.c_uint8 * 128)()
        if self.rank == 0:
            _lib.check(lib.mrs_exchange_unique_id(ident))
        if self.world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        self._h = C.c_void_p()
        _lib.check(lib.mrs_exchange_create(_lib.ctx(self.device), self.world, self.rank, ident, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib_mod.load().mrs_exchange_destroy(self._h)
        except Exception:
            pass

    @staticmethod
    def _bytes(t):
        return torch.view_as_real(t) if t.is_complex() else t

    def allgather(self, local):
        _lib = self._lib_mod
        x = local.contiguous()
        assert x.is_cuda and x.shape[0] > 0
        out = torch.empty((self.world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        entry = x[0].numel() * x.element_size()
        _lib.check(_lib.load().mrs_exchange_allgather(self._h, _lib.ptr(self._bytes(x)), self._C.c_int64(x.shape[0]), self._C.c_int64(entry),
                                                      _lib.ptr(self._bytes(out)), _lib.current_stream(self.device)))
        return out

    def allgather_into(self, out, local, stream=None):
        """ONE ncclAllGather of `local` (contiguous, the same size on every rank) into `out` (world x local, rank order), enqueued on `stream`
        (a torch stream; default: the current one): stream-ordered, nothing blocks the host"""
        _lib = self._lib_mod
        assert local.is_cuda and out.is_cuda and local.is_contiguous() and out.is_contiguous()
        nbytes = local.numel() * local.element_size()
        assert out.numel() * out.element_size() == self.world * nbytes
        st = self._C.c_void_p(stream.cuda_stream) if stream is not None else _lib.current_stream(self.device)
        _lib.check(_lib.load().mrs_exchange_allgather(self._h, self._C.c_void_p(local.data_ptr()), self._C.c_int64(1), self._C.c_int64(nbytes),
                                                      self._C.c_void_p(out.data_ptr()), st))
        return out

    def fetch_plan(self, global_rows, rows_per_rank):
        """-> PlannedFetch: the request phase of fetch_rows done now (collective: every rank calls it), fetches are then stream-ordered"""
        return PlannedFetch(self, global_rows, rows_per_rank)

    def fetch_rows(self, local_db, global_rows):
        _lib = self._lib_mod
        db = local_db.contiguous()
        rows = global_rows.to(torch.int64).reshape(-1).contiguous()
        assert db.is_cuda and rows.is_cuda and rows.numel() > 0
        entry = db[0].numel() * db.element_size()
        out = torch.empty((rows.numel(),) + tuple(db.shape[1:]), dtype=db.dtype, device=db.device)
        _lib.check(_lib.load().mrs_exchange_fetch_rows(self._h, _lib.ptr(self._bytes(db)), self._C.c_int64(db.shape[0]), self._C.c_int64(entry),
                                                       _lib.ptr(rows), int(rows.numel()), _lib.ptr(self._bytes(out)), _lib.current_stream(self.device)))
        return out


class PlannedFetch:
    """RowFetchPlan's counterpart behind the C ABI (mrs_exchange_fetch_plan_*): the candidate rows a rank will ask for are registered once
    (one all-gather of the requests, one host synchronisation), every fetch() afterwards is gather -> grouped RCCL send / receive -> scatter,
    enqueued on a stream without touching the host.  Same interface as RowFetchPlan.fetch(async_op=True): (work, finish)."""

    class _Work:
        def __init__(self, event):
            self.event = event

        def wait(self):
            torch.cuda.current_stream().wait_event(self.event)       # stream-side wait, like a torch.distributed Work on a CUDA stream

    def __init__(self, exchange, global_rows, rows_per_rank):
        import ctypes as C
        self.x, self._C = exchange, C
        _lib = exchange._lib_mod
        rows = global_rows.to(torch.int64).reshape(-1).contiguous()
        assert rows.is_cuda and rows.numel() > 0
        self.n, self.rows_per_rank = int(rows.numel()), int(rows_per_rank)
        self._h = C.c_void_p()
        _lib.check(_lib.load().mrs_exchange_fetch_plan_create(exchange._h, C.c_int64(self.rows_per_rank), _lib.ptr(rows), self.n,
                                                              _lib.current_stream(exchange.device), C.byref(self._h)))
        sent, recv = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().mrs_exchange_fetch_plan_counts(self._h, C.byref(sent), C.byref(recv)))
        self.rows_from_peers, self.rows_to_peers = int(recv.value), int(sent.value)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.x._lib_mod.load().mrs_exchange_fetch_plan_destroy(self._h)
        except Exception:
            pass

    def bytes_in(self, row_bytes, rank_local_too=False):
        return row_bytes * (self.n if rank_local_too else self.rows_from_peers)

    def fetch(self, local_db, async_op=False, stream=None):
        """rows in request order; async_op: (work, finish) -- the fetch runs on `stream` (a torch stream the caller keeps for communication;
        it first waits for what the current stream has enqueued so far), work.wait() makes the then-current stream wait for it"""
        _lib = self.x._lib_mod
        db = local_db.contiguous()
        assert db.is_cuda and db.shape[0] == self.rows_per_rank
        real = torch.view_as_real(db) if db.is_complex() else db
        entry = real[0].numel() * real.element_size()
        out = torch.empty((self.n,) + tuple(db.shape[1:]), dtype=db.dtype, device=db.device)
        oreal = torch.view_as_real(out) if out.is_complex() else out
        cur = torch.cuda.current_stream()
        st = stream if stream is not None else cur
        if st is not cur:
            ready = torch.cuda.Event(); ready.record(cur)
            st.wait_event(ready)
        _lib.check(_lib.load().mrs_exchange_fetch_planned(self._h, _lib.ptr(real), self._C.c_int64(entry), _lib.ptr(oreal), self._C.c_void_p(st.cuda_stream)))
        if not async_op:
            if st is not cur:
                done = torch.cuda.Event(); done.record(st); cur.wait_event(done)
            return out
        done = torch.cuda.Event(); done.record(st)
        self._keep = (db, out)
        return PlannedFetch._Work(done), (lambda: out)
