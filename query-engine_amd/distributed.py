"""Exchange step of the multi-GPU aggregate: every rank contributes the small table its PARTIAL
aggregate produced (keys, then partial sums / counts / minima / maxima) and receives the tables of all
ranks, ready to be merged by a FINAL aggregate.  The row-parallel part of the plan (scan, selection,
projection, partial aggregate) needs no communication at all; this is the only collective.

Tables are tiny next to the scanned data (groups x columns x 8 bytes), so the exchange is one
fixed-size all-gather (an int64 payload that carries the row count and every column, float64 columns
as their bit patterns) into preallocated buffers: no size negotiation, no allocation per step, and
the same code runs over RCCL (`nccl` backend, device tensors) and over gloo (CPU tensors, used by
the tests)."""


class PartialExchange:
    def __init__(self, torch, dist, device, world, max_groups, dtypes):
        """dtypes: per partial-table column, torch.int64 or torch.float64."""
        self.torch, self.dist, self.world, self.max_groups = torch, dist, world, max_groups
        self.dtypes = list(dtypes)
        for d in self.dtypes:
            if d not in (torch.int64, torch.float64):
                raise ValueError("partial tables hold int64 and float64 columns only")
        # One int64 payload per rank: [row count | column 0 | column 1 ...], each column max_groups long;
        # float64 columns travel as their bit patterns.  Rows past the count are zero.
        self.n_cols = len(self.dtypes)
        self.len = 1 + self.n_cols * max_groups
        self.send = torch.zeros(self.len, dtype=torch.int64, device=device)
        self.recv = torch.zeros(world * self.len, dtype=torch.int64, device=device)
        self.sent_rows = 0

    def _gather(self, columns, with_count):
        torch = self.torch
        g = int(columns[0].shape[0]) if columns else 0
        if g > self.max_groups:
            raise RuntimeError(f"{g} groups do not fit the exchange buffer ({self.max_groups}); size it from the key domain")
        if len(columns) != self.n_cols:
            raise ValueError("column count differs from the exchange layout")
        body = self.send[1:].view(self.n_cols, self.max_groups)
        if g < self.sent_rows:
            body[:, g:self.sent_rows].zero_()  # a shorter table than last time: clear what it leaves behind
        self.sent_rows = g
        for i, c in enumerate(columns):
            body[i, :g].copy_(c.view(torch.int64) if c.dtype == torch.float64 else c)
        if with_count:
            self.send[0] = g
        self.dist.all_gather_into_tensor(self.recv, self.send)

    def _typed(self, t, i):
        return t.view(self.torch.float64) if self.dtypes[i] == self.torch.float64 else t

    def exchange(self, columns):
        """columns: this rank's partial table, one 1-D tensor per column, equal lengths.
        Returns, per rank, the list of column tensors (views into the receive buffer: valid until
        the next call).  Reads the row counts back to the host (one synchronisation)."""
        self._gather(columns, True)
        per_rank = self.recv.view(self.world, self.len)
        counts = per_rank[:, 0].tolist()
        tables = []
        for r in range(self.world):
            body = per_rank[r, 1:].view(self.n_cols, self.max_groups)
            tables.append([self._typed(body[i, :int(counts[r])], i) for i in range(self.n_cols)])
        return tables

    def exchange_merged(self, columns):
        """Same exchange, without the trip to the host: returns one table of world x max_groups rows
        (list of contiguous column tensors) in which the rows no rank filled are all zero.  A partial
        table with a COUNT(*) column tells the two apart (count > 0), so the caller puts that
        predicate under its FINAL aggregate and merges everything in one launch."""
        self._gather(columns, False)
        body = self.recv.view(self.world, self.len)[:, 1:].view(self.world, self.n_cols, self.max_groups)
        merged = body.permute(1, 0, 2).reshape(self.n_cols, self.world * self.max_groups)  # one small copy
        return [self._typed(merged[i], i) for i in range(self.n_cols)]


class BlockExchange:
    """The exchange for partial tables that already lie in one device block (an aggregate run with
    deferred group count returns all its columns as slices of one allocation, one row per table slot,
    unused rows zero): the block is the all-gather payload as it is (one copy into the send buffer
    instead of one per column), and one gather kernel turns the received blocks into a single table of world x rows rows whose columns
    are contiguous.  Column i of the block starts `offsets[i]` 8-byte words into it.  As with
    PartialExchange.exchange_merged, COUNT(*) > 0 tells real rows from padding."""

    def __init__(self, torch, dist, device, world, words, offsets, rows, dtypes):
        self.torch, self.dist, self.device, self.world = torch, dist, device, world
        self.words, self.rows, self.dtypes = words, rows, list(dtypes)
        self.send = torch.zeros(words, dtype=torch.int64, device=device)
        self.recv = torch.zeros(world * words, dtype=torch.int64, device=device)
        r = torch.arange(world, dtype=torch.int64).view(1, world, 1) * words
        j = torch.arange(rows, dtype=torch.int64).view(1, 1, rows)
        o = torch.tensor(list(offsets), dtype=torch.int64).view(len(offsets), 1, 1)
        self.index = (r + j + o).reshape(-1).to(device)  # [column][rank][row] -> word in recv

    def exchange(self, block):
        """block: int64 tensor over this rank's partial-table block (`words` long)."""
        if block.shape[0] != self.words:
            raise ValueError("block size differs from the exchange layout")
        # one copy into a buffer the collective library has seen before: the block is engine memory, a different
        # allocation every step
        self.send.copy_(block)
        self.dist.all_gather_into_tensor(self.recv, self.send)
        merged = self.recv[self.index].view(len(self.dtypes), self.world * self.rows)
        return [merged[i].view(self.torch.float64) if d == self.torch.float64 else merged[i] for i, d in enumerate(self.dtypes)]


def shard_range(total_rows, rank, world):
    """Contiguous row range of `rank`: concatenating the ranges in rank order gives the table."""
    return rank * total_rows // world, (rank + 1) * total_rows // world


class KeyPartitionedExchange:
    """Exchange for partial tables too large to replicate (about a million groups): every partial row
    goes to the rank that owns its key (hash of the first column modulo the number of ranks) with one
    all-to-all, each rank merges only its share of the key space, and the result is the concatenation
    of the ranks' outputs.  Per rank (W - 1) / W of its table crosses the links once, spread over all
    peers at the same time (xGMI is point to point: W - 1 direct transfers of 1 / W each), instead of
    W tables arriving at every rank as in an all-gather.

    Columns are int64 or float64; they travel as one [rows, columns] int64 matrix (float64 bit patterns
    reinterpreted), so a step costs one small all-to-all of row counts and one of payload."""

    def __init__(self, torch, dist, device, world, dtypes):
        self.torch, self.dist, self.device, self.world = torch, dist, device, world
        self.dtypes = list(dtypes)
        for d in self.dtypes:
            if d not in (torch.int64, torch.float64):
                raise ValueError("partial tables hold int64 and float64 columns only")
        self.send_counts = torch.zeros(world, dtype=torch.int64, device=device)
        self.recv_counts = torch.zeros(world, dtype=torch.int64, device=device)

    def owner(self, keys):
        """Rank that merges each key (int64 tensor): multiplicative hash, high bits, modulo ranks."""
        h = (keys * -7046029254386353131) >> 17  # wraps; arithmetic shift
        return (h & 0x7FFFFFFF) % self.world

    def exchange(self, columns):
        """columns: this rank's partial table (column 0 = int64 key).  Returns the rows this rank owns,
        from all ranks, as a list of column tensors (freshly allocated)."""
        torch = self.torch
        if len(columns) != len(self.dtypes):
            raise ValueError("column count differs from the exchange layout")
        rows = int(columns[0].shape[0])
        payload = torch.empty((rows, len(columns)), dtype=torch.int64, device=self.device)
        for i, c in enumerate(columns):
            payload[:, i] = c.to(self.device).view(torch.int64) if c.dtype == torch.float64 else c.to(self.device)
        dest = self.owner(payload[:, 0])
        order = torch.argsort(dest, stable=True)
        payload = payload[order].contiguous()
        self.send_counts.copy_(torch.bincount(dest, minlength=self.world))
        self.dist.all_to_all_single(self.recv_counts, self.send_counts)
        send = self.send_counts.tolist()
        recv = self.recv_counts.tolist()
        out = torch.empty((sum(recv), len(columns)), dtype=torch.int64, device=self.device)
        self.dist.all_to_all_single(out, payload, output_split_sizes=recv, input_split_sizes=send)
        return [out[:, i].contiguous().view(torch.float64) if d == torch.float64 else out[:, i].contiguous()
                for i, d in enumerate(self.dtypes)]


def make_comm(qe, ctx, dist, rank, world):
    """A qe.Comm (the library's own RCCL communicator: qe_comm_init) for this rank.  The 128-byte rendezvous id is made
    by rank 0 and handed to the others through torch.distributed, whatever its backend: the exchanges themselves
    (qe_exchange_all_gather / qe_exchange_by_key) never touch torch.  Before any rank enters the collective
    qe_comm_init, every rank confirms that it can bind RCCL and the ranks agree on that through torch.distributed: a
    rank that could not would otherwise leave the others waiting inside the rendezvous.  Raises qe.QeError on every
    rank when any rank cannot take part."""
    failure = None
    try:
        qe.Comm.library_rccl_version()
    except qe.QeError as e:
        failure = e
    ids = [None]
    if rank == 0 and failure is None:
        try:
            ids[0] = qe.Comm.unique_id()
        except qe.QeError as e:
            failure = e
    if world > 1:
        able = [None] * world
        dist.all_gather_object(able, failure is None)
        if not all(able):
            raise failure if failure is not None else qe.QeError(41)
        dist.broadcast_object_list(ids, src=0)
    if ids[0] is None:
        raise failure if failure is not None else qe.QeError(41)
    return qe.Comm(ctx, rank, world, ids[0])
