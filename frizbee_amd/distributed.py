"""Multi-GPU composition of the hot path: the haystack list is sharded by contiguous index range, one process
per GPU scores its shard with a global `index_offset` (exactly what `match_list_parallel`'s workers do with
2048-item chunks, reference src/matcher/parallel.rs:55-63), then the per-shard, index-ordered match lists are
exchanged with ONE gather / all-gather (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests) and combined into the list
the reference's per-run sort + k-way merge returns (parallel.rs:66-87).  With the runs in the root's HBM that is ONE device pass
(`ShardExchange.collect_merged` -> fzb_merge_shard_runs: shard order is ascending index order, so the concatenation is the list
`match_list` orders - reverse / stable radix sort once, one copy to the host); `merge_shard_runs` is the host form of the same combine
(CPU tensors, the gloo tests).  Scoring itself needs no collective."""
import numpy as np
import torch
import torch.distributed as dist

import ctypes as C

from . import MATCH_DTYPE, SortStrategy, _check, _take, k_merge_matches, lib, radix_sort_matches


def shard_range(n_total, rank, world):
    """Contiguous index range [lo, hi) of shard `rank` (g*ceil(N/G) .. min((g+1)*ceil(N/G), N))."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def shard_ranges_by_bytes(ends, world):
    """Byte-balanced contiguous shards of a ragged list (BASELINE config 4): `ends` = exclusive byte end of every haystack (uint64,
    non-decreasing).  Shard g is the index range [cut[g], cut[g+1]) where cut[g] is the first haystack that STARTS at or after
    g/world of the total bytes (a haystack that straddles the byte target stays with the shard it starts in) - contiguous and in
    list order like `shard_range`, so `index_offset = cut[g]` keeps indices global (what match_list_parallel's workers do with their
    chunk starts, src/matcher/parallel.rs:55-63), but every GPU streams about the same number of bytes.  Returns the list of (lo, hi).
    The C ABI's fzb_shard_ranges(by_bytes = 1) applies the same rule (tests/test_host_abi.py compares the two)."""
    ends = np.asarray(ends, dtype=np.uint64)
    n = len(ends)
    total = int(ends[-1]) if n else 0
    cuts = [0]
    for g in range(1, world):
        # haystack i starts at ends[i-1]: the first i with ends[i-1] >= target is one past the first END that reaches the target
        # (ends 10,20,30,40 and world 2: target 20 -> cut 2, shards [0,2) and [2,4), 20 bytes each)
        target = (total * g) // world
        cuts.append(int(np.searchsorted(ends, np.uint64(target), side="left")) + 1 if target else 0)
    cuts.append(n)
    cuts = [min(max(c, cuts[i - 1] if i else 0), n) for i, c in enumerate(cuts)]
    return [(cuts[g], cuts[g + 1]) for g in range(world)]


def all_gather_matches(records_u8, count, group=None):
    """records_u8: uint8 tensor holding >= count 8-byte records (device or CPU); count: python int or 0-dim/1-elem int tensor.
    Returns a list (one per rank) of numpy MATCH_DTYPE arrays.  Two collectives: counts, then records padded to the max count
    (all-gather-v emulation)."""
    world = dist.get_world_size(group)
    dev = records_u8.device
    cnt = count.to(torch.int64).reshape(1) if torch.is_tensor(count) else torch.tensor([int(count)], dtype=torch.int64, device=dev)
    counts = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, cnt.to(dev), group=group)
    counts_h = counts.cpu().tolist()
    mx = max(max(counts_h), 1)
    send = records_u8.reshape(-1)[: mx * 8]
    if send.numel() < mx * 8:  # local buffer shorter than the largest shard's result: pad
        pad = torch.zeros(mx * 8, dtype=torch.uint8, device=dev)
        pad[: send.numel()] = send
        send = pad
    recv = torch.empty(world * mx * 8, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    host = recv.cpu().numpy().reshape(world, mx * 8)
    return [host[r, : counts_h[r] * 8].copy().view(MATCH_DTYPE) for r in range(world)]


class ShardExchange:
    """Steady-state exchange of per-shard match lists with NO host synchronisation in the loop: a gather to the root
    rank (the process whose host consumes the merged result) of fixed-capacity buffers, double-buffered and asynchronous.

    Buffer layout per rank and slot: [u32 records written | u32 matches found | capacity x 8-byte records].  The scoring pipeline writes the
    count and the records straight into it (`count_ptr` / `records_ptr` are what `fzb_match_list_device` takes), `post`
    starts the transport on the backend's own stream (RCCL send/recv over the point-to-point xGMI links: the root
    receives its world-1 peers' buffers on separate links in parallel), and the next step's kernels overlap it.
    The capacity is agreed once, up front (`plan`), from a first measured count; a shard that later outgrows it is
    reported by `collect`, never truncated silently."""

    HEADER = 8

    def __init__(self, capacity_records, device, group=None, root=0, slots=2, host_staged=None, transport=None):
        # transport "p2p" (default): every other rank sends its buffer to the root, the root posts one receive per peer, batched into ONE RCCL
        # group call (dist.batch_isend_irecv) - the root's own run never travels: its receive slot IS its send buffer, and with one rank
        # nothing is posted at all.  "gather": dist.gather of the whole list (what rounds 1-5 ran: the same sends and receives plus a
        # device-to-device copy of the root's own 4 MB on RCCL's stream and its two cross-stream event waits - 18 us per step with one rank).
        # FZB_EXCHANGE_TRANSPORT overrides the default.
        import os
        self.transport = transport or os.environ.get("FZB_EXCHANGE_TRANSPORT", "p2p")
        if self.transport not in ("p2p", "gather"):
            raise ValueError(f"ShardExchange transport {self.transport!r}: 'p2p' or 'gather'")
        self.group, self.root = group, root
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device(device)
        self.slots = slots
        backend = dist.get_backend(group)
        # A backend without device collectives (gloo: the rank-count rehearsal on one GPU, `FZB_BENCH_BACKEND=gloo bench.py --gpus 2`) moves
        # CPU tensors: the pipeline still writes into a device buffer, `post` copies it to the host first (a synchronising copy) and the
        # root combines on the host.  With RCCL nothing is staged.
        self.host_staged = (self.device.type == "cuda" and backend != "nccl") if host_staged is None else bool(host_staged)
        # only RCCL's completion query is trusted to stand in for wait(): on other backends wait() is what surfaces a failed collective
        self._skip_completed_wait = backend == "nccl"
        self.grown = 0  # how often `grow` re-sized the exchange (ordered_query)
        self._alloc(int(capacity_records))

    def _alloc(self, capacity_records):
        self.cap = int(capacity_records)
        nbytes = self.HEADER + self.cap * 8
        xdev = torch.device("cpu") if self.host_staged else self.device
        self.send = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(self.slots)]
        self.send_x = [torch.zeros(nbytes, dtype=torch.uint8, device=xdev) for _ in range(self.slots)] if self.host_staged else self.send
        self.recv = [[torch.zeros(nbytes, dtype=torch.uint8, device=xdev) for _ in range(self.world)] if self.rank == self.root else None for _ in range(self.slots)]
        if self.rank == self.root and self.transport == "p2p":
            for s_ in range(self.slots):
                self.recv[s_][self.root] = self.send_x[s_]  # the root's own run: in place
        self.work = [None] * self.slots
        self._merge_args = [None] * self.slots

    def grow(self, capacity_records):
        """Every rank, together: re-size the exchange (all pending exchanges are drained first).  The buffers move: cached addresses are dropped."""
        for s in range(self.slots):
            self.wait(s)
        self._alloc(max(int(capacity_records), self.cap))
        self.grown += 1

    @staticmethod
    def plan(local_count, group=None, margin=1.25, device=None):
        """Capacity every rank agrees on: max over ranks of the measured count, plus a margin (one collective, set-up only)."""
        t = torch.tensor([int(local_count)], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return int(int(t.item()) * margin) + 4096

    def count_ptr(self, slot):
        return self.send[slot].data_ptr()

    def records_ptr(self, slot):
        return self.send[slot].data_ptr() + self.HEADER

    def wait(self, slot):
        """Order the caller's stream after the exchange that last used `slot` (no host block with RCCL).  An exchange that has already
        completed - the steady state: it is two steps old - needs no ordering at all, and skipping the stream-level wait keeps a barrier
        packet out of the scoring stream and 10 us of host time out of the step (tools/exp_dist_overhead.py)."""
        works = self.work[slot]
        if works is not None:
            for w in works:
                done = False
                if self._skip_completed_wait:
                    try:
                        done = bool(w.is_completed())
                    except Exception:  # no completion query: order the streams
                        done = False
                if not done:
                    w.wait()  # (raises what the collective raised)
            self.work[slot] = None

    def post(self, slot, stream=None):
        """Start moving this slot's buffer to the root (asynchronous).  `stream` = the raw HIP stream the pipeline was enqueued on when that is
        not torch's current stream: the exchange (and the host-staged copy in front of it) is ordered behind it."""
        self.wait(slot)
        if stream is not None and self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            if int(stream) != int(cur.cuda_stream):  # (the usual case - the pipeline's stream IS torch's current stream - costs nothing)
                cur.wait_stream(torch.cuda.ExternalStream(int(stream), device=self.device))
        if self.host_staged:
            self.send_x[slot].copy_(self.send[slot])  # device -> host, behind the pipeline on the current stream (synchronises)
        if self.transport == "gather":
            self.work[slot] = [dist.gather(self.send_x[slot], gather_list=self.recv[slot], dst=self.root, group=self.group, async_op=True)]
            return
        if self.rank == self.root:
            ops = [dist.P2POp(dist.irecv, self.recv[slot][r], r, self.group) for r in range(self.world) if r != self.root]
        else:
            ops = [dist.P2POp(dist.isend, self.send_x[slot], self.root, self.group)]
        self.work[slot] = dist.batch_isend_irecv(ops) if ops else None

    def bytes_per_rank(self):
        """what one rank ships per exchange: header + the whole fixed-capacity record buffer"""
        return self.HEADER + self.cap * 8

    def max_found(self, slot):
        """Root only, after wait(slot): the largest `matches found` any shard reported in this exchange (what the capacity has to hold)."""
        hdr = torch.stack([b[: self.HEADER] for b in self.recv[slot]]).cpu().numpy().view(np.uint32).reshape(self.world, 2)
        return int(hdr.max())

    def ordered_query(self, run, matcher, slot=0, stream=None, copy=False, max_retries=6):
        """One query, every rank together, the ordered list on the root (None elsewhere) - `match_list_parallel`'s result
        (src/matcher/parallel.rs:18-89) whatever the number of matches: `run(records_ptr, capacity, count_ptr)` enqueues this rank's
        pipeline (fzb_match_list_device with its global index_offset) into the exchange buffer; the runs are gathered to the root and
        combined there (`collect_merged`); then the root tells every rank, with ONE 8-byte broadcast, whether every shard's run fitted
        the exchange.  If one did not (its `matches found` exceeds the capacity: the capacity was planned from an earlier query), every rank
        re-sizes the exchange to 1.25 x the largest run and the query is repeated - a second query with more matches than the first can
        never fail with FZB_ERR_CAPACITY or return a shorter list."""
        ctl_dev = torch.device("cpu") if (self.host_staged or self.device.type != "cuda") else self.device
        for _ in range(max_retries + 1):
            self.wait(slot)
            run(self.records_ptr(slot), self.cap, self.count_ptr(slot))
            self.post(slot)
            merged, need, err = None, 0, None
            if self.rank == self.root:
                try:  # (a truncated run makes the combine fail - FZB_ERR_CAPACITY on the device, RuntimeError on the host - and only then are the headers read)
                    merged = self.collect_merged(slot, matcher, stream=stream, copy=copy)
                except Exception as e:  # noqa: BLE001 - anything else is re-raised below, after the other ranks have been told
                    found = self.max_found(slot)
                    if found > self.cap:
                        need = found
                    else:
                        need, err = -1, e
            else:
                self.wait(slot)
            t = torch.tensor([need], dtype=torch.int64, device=ctl_dev)
            dist.broadcast(t, src=self.root, group=self.group)
            need = int(t.item())
            if need == 0:
                return merged
            if need < 0:
                raise err if err is not None else RuntimeError("ordered_query: the root rank failed to combine the runs")
            self.grow(int(need * 1.25) + 4096)
        raise RuntimeError("ordered_query: the exchange kept overflowing (a shard's match count grows between retries?)")

    def collect(self, slot):
        """Root only, synchronising: the per-rank runs of the exchange posted on `slot` as numpy MATCH_DTYPE arrays."""
        self.wait(slot)
        if self.rank != self.root:
            return None
        runs = []
        for r, (cnt, total) in enumerate(self._headers(slot)):
            buf = self.recv[slot][r]
            runs.append(buf[self.HEADER : self.HEADER + cnt * 8].cpu().numpy().copy().view(MATCH_DTYPE))  # the records that exist, not the capacity
        return runs

    def _headers(self, slot):
        """[(records written, matches found)] per rank - fzb_match_list_device's two counters; raises when a shard outgrew the capacity"""
        hdr = torch.stack([b[: self.HEADER] for b in self.recv[slot]]).cpu().numpy().view(np.uint32).reshape(self.world, 2)
        out = []
        for r in range(self.world):
            cnt, total = int(hdr[r, 0]), int(hdr[r, 1])
            if total > self.cap or cnt > self.cap:
                raise RuntimeError(f"shard {r} produced {max(total, cnt)} matches, exchange capacity is {self.cap}: plan a larger capacity")
            out.append((cnt, total))
        return out

    def collect_merged(self, slot, matcher, stream=None, copy=False):
        """Root only, synchronising: the exchange posted on `slot` as ONE list in `matcher.config.sort` order - what
        `match_list_parallel` returns (parallel.rs:66-87).  The gathered runs never leave the root's HBM unordered: concatenation in rank
        order (= ascending index order) + reverse / stable radix sort on the device (fzb_merge_shard_runs), one copy to the host.
        CPU tensors (the gloo tests) take the host form, `merge_shard_runs(collect(slot), sort)`."""
        self.wait(slot)
        if self.rank != self.root:
            return None
        bufs = self.recv[slot]
        if stream is None and bufs[0].is_cuda:
            # wait() ordered TORCH'S CURRENT stream behind the gather: the concatenation must be launched on that stream (the legacy null
            # stream is not ordered after the collective when the caller's current stream is a non-blocking one)
            stream = torch.cuda.current_stream(bufs[0].device).cuda_stream
        if not bufs[0].is_cuda:
            return merge_shard_runs(self.collect(slot), matcher.config.sort)
        if self._merge_args[slot] is None:  # the buffers never move: their addresses are marshalled once
            self._merge_args[slot] = matcher.merge_args([b.data_ptr() + self.HEADER for b in bufs], [b.data_ptr() for b in bufs], [self.cap] * self.world)
        # a shard that outgrew the exchange capacity makes the call fail (FZB_ERR_CAPACITY): reported, never returned as a shorter list
        # (copy=False: the array wraps the library's pinned buffer, which returns to the pool when the array is collected - a fresh numpy
        # copy of a 4 MB result costs more than the device-side merge)
        return matcher.merge_shard_runs(*self._merge_args[slot], stream=stream, copy=copy)


def merge_shard_runs(runs, sort):
    """Per-shard index-ordered runs -> the reference's final ordering (parallel.rs:66-87)."""
    sort = SortStrategy(int(sort))
    prepared = []
    for r in runs:
        r = np.ascontiguousarray(r)
        if sort in (SortStrategy.IndexDesc, SortStrategy.ScoreThenIndexDesc):
            r = r[::-1].copy()
        if sort in (SortStrategy.ScoreThenIndexAsc, SortStrategy.ScoreThenIndexDesc):
            r = radix_sort_matches(r)
        prepared.append(r)
    return k_merge_matches(sort, prepared)


class RcclShardComm:
    """The multi-process form BELOW the C ABI (fzb_shard_comm, csrc/host_rccl.hip): one process per GPU, the runs exchanged by RCCL
    inside libfrizbee_hip.so itself - what a Rust host binds ("One process per GPU, below the boundary"); torch.distributed is used here only to carry the
    128-byte communicator id from rank 0 to the other ranks (any channel does).  `match_list_parallel(matcher, shard, index_offset)` is
    `Matcher::match_list_parallel` (src/matcher/parallel.rs:18-89) over the WHOLE list: the ordered result on rank 0 (every rank with
    all_ranks=True), an empty array elsewhere."""

    ID_BYTES = 128

    def __init__(self, rank=None, world=None, unique_id=None, group=None):
        if rank is None or world is None:
            rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
        if unique_id is None:
            unique_id = self.unique_id() if rank == 0 else bytes(self.ID_BYTES)
            if world > 1:
                # the one out-of-band step: rank 0's id to everyone (a CPU tensor for gloo, a device tensor for nccl)
                on_dev = dist.get_backend(group) == "nccl"
                t = torch.frombuffer(bytearray(unique_id), dtype=torch.uint8)
                t = t.cuda() if on_dev else t
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                unique_id = bytes(t.cpu().numpy().tobytes())
        if len(unique_id) != self.ID_BYTES:
            raise ValueError(f"the communicator id has {self.ID_BYTES} bytes")
        self.h = C.c_void_p()
        _check(lib().fzb_shard_comm_create(C.c_char_p(bytes(unique_id)), int(rank), int(world), C.byref(self.h)))  # collective
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id():
        """fzb_rccl_unique_id: 128 opaque bytes, drawn on rank 0."""
        buf = (C.c_uint8 * RcclShardComm.ID_BYTES)()
        _check(lib().fzb_rccl_unique_id(buf))
        return bytes(buf)

    def match_list_parallel(self, matcher, shard, index_offset, all_ranks=False, copy=True):
        """fzb_match_list_parallel_rccl (collective): `shard` = this rank's resident Corpus, `index_offset` = its first global index."""
        out, n = C.c_void_p(), C.c_size_t()
        _check(lib().fzb_match_list_parallel_rccl(matcher.h, shard.h, int(index_offset), self.h, 1 if all_ranks else 0, C.byref(out), C.byref(n)))
        return _take(out, n, copy)

    def last_exchange_bytes(self):
        """(sent, received) record bytes of this rank's last query."""
        b = (C.c_uint64 * 2)()
        _check(lib().fzb_shard_comm_last_exchange(self.h, b))
        return int(b[0]), int(b[1])

    def close(self):
        if self.h:
            lib().fzb_shard_comm_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

