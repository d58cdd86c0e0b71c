"""Multi-GPU layout of the hot path: one process per GPU, RecordBatches sharded by contiguous row
ranges (every reference kernel is per-chunk independent: src/functions/scalar.rs:28-31,
src/table.rs:98-105, src/functions/aggregate.rs:88-90), no data-path collective.  The only exchange is
the combine of the per-rank {sum, min, max, count} partials, folded in rank order so the f64 result
is deterministic.  Works over any torch.distributed backend ("nccl" = RCCL on the GPU box, "gloo" in
the CPU tests).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

from ._abi import F32, F64, AggResult


def shard_rows(total_rows: int, world: int, rank: int, align: int = 1024) -> Tuple[int, int]:
    """Contiguous row range [begin, end) of `rank`; boundaries fall on `align`-row batch edges
    (1024 = the reference readers' RecordBatch size, src/dataframe.rs:352)."""
    nb = (total_rows + align - 1) // align
    b0 = nb * rank // world
    b1 = nb * (rank + 1) // world
    return min(b0 * align, total_rows), min(b1 * align, total_rows)


def shard_chunks(chunk_lens: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Whole RecordBatches per rank: [first_chunk, last_chunk) per rank, balanced by rows, order kept
    (chunk order = rank order preserves row order of sharded outputs)."""
    total = sum(chunk_lens)
    bounds, acc, c = [], 0, 0
    for r in range(world):
        first = c
        target = total * (r + 1) / world
        while c < len(chunk_lens) and (acc + chunk_lens[c] / 2.0 <= target or r == world - 1):
            acc += chunk_lens[c]
            c += 1
        bounds.append((first, c))
    return bounds


def combine(parts: Sequence[AggResult]) -> AggResult:
    """Fold per-rank partial aggregates in rank order (AggregateFunctions' own chunk fold,
    src/functions/aggregate.rs:82-93, with ranks in place of chunks)."""
    dt = parts[0].dtype
    is_f = dt in (F32, F64)
    tot = 0.0 if is_f else 0
    mn = mx = None
    cnt = 0
    for p in parts:
        tot = tot + p.sum
        cnt += p.count
        if p.is_some:
            if is_f:
                # NaN-ignoring like the device fold: a NaN partial never displaces a number
                mn = p.min if mn is None or (p.min < mn) or (mn != mn) else mn
                mx = p.max if mx is None or (p.max > mx) or (mx != mx) else mx
            else:
                mn = p.min if mn is None else min(mn, p.min)
                mx = p.max if mx is None else max(mx, p.max)
    if not is_f:
        bits = {0: 8, 1: 16, 2: 32, 3: 64, 4: 8, 5: 16, 6: 32, 7: 64, 10: 64}[dt]
        tot &= (1 << bits) - 1
        if dt <= 3 and tot >= 1 << (bits - 1):
            tot -= 1 << bits
    return AggResult(tot, mn if mn is not None else 0, mx if mx is not None else 0, cnt, cnt > 0, dt)


_gather_buffers = {}

# A 1-rank process group normally short-circuits every combine.  bench.py --force-exchange (and the -m gpu nccl test) set
# this so that the collectives themselves run on a 1-rank RCCL communicator: the N > 1 code path, one GPU.
FORCE_COLLECTIVES = False


def _single(dist) -> bool:
    return not dist.is_initialized() or (dist.get_world_size() == 1 and not FORCE_COLLECTIVES)


def _all_gather_words(words, device):
    """One collective for a small int64 vector: returns [world][len(words)] as Python ints.  Buffers are cached per
    (length, device) so the per-step cost is one all_gather and one D2H."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    key = (len(words), str(device))
    if key not in _gather_buffers:
        dev = device if device is not None else "cpu"
        _gather_buffers[key] = (torch.zeros(len(words), dtype=torch.int64, device=dev), torch.zeros(world * len(words), dtype=torch.int64, device=dev))
    inp, out = _gather_buffers[key]
    inp.copy_(torch.tensor(words, dtype=torch.int64))
    try:
        dist.all_gather_into_tensor(out, inp)
    except (RuntimeError, NotImplementedError, AttributeError):   # a backend without the flat variant
        parts = [torch.zeros_like(inp) for _ in range(world)]
        dist.all_gather(parts, inp)
        out = torch.cat(parts)
    flat = out.tolist()
    n = len(words)
    return [flat[r * n:(r + 1) * n] for r in range(world)]


def _f64_bits(x: float) -> int:
    import struct
    return struct.unpack("<q", struct.pack("<d", float(x)))[0]


def _bits_f64(b: int) -> float:
    import struct
    return struct.unpack("<d", struct.pack("<q", int(b)))[0]


def all_combine(local: Sequence[AggResult], device=None) -> List[AggResult]:
    """all_gather every rank's partials (ONE collective of 5 int64 words per value: f64 partials travel as their bit
    patterns, integer partials as two's-complement i64 — both exact) and fold them identically on every rank."""
    import torch.distributed as dist
    if _single(dist):
        return [combine([p]) for p in local]
    world = dist.get_world_size()
    wrap = lambda x: x - (1 << 64) if x >= (1 << 63) else x
    words = []
    for p in local:
        if p.dtype in (F32, F64):
            words += [_f64_bits(p.sum), _f64_bits(p.min), _f64_bits(p.max)]
        else:
            words += [wrap(int(p.sum)), wrap(int(p.min)), wrap(int(p.max))]
        words += [int(p.count), int(p.is_some)]
    gathered = _all_gather_words(words, device)
    out = []
    for v, p in enumerate(local):
        parts = []
        for r in range(world):
            s, a, b, cnt, some = gathered[r][5 * v:5 * v + 5]
            if p.dtype in (F32, F64):
                s, a, b = _bits_f64(s), _bits_f64(a), _bits_f64(b)
            elif p.dtype == 7:  # U64 travels as i64
                s, a, b = [x + (1 << 64) if x < 0 else x for x in (s, a, b)]
            parts.append(AggResult(s, a, b, int(cnt), bool(some), p.dtype))
        out.append(combine(parts))
    return out


def combine_groups(parts):
    """Fold per-rank results of Api.group_pipeline — (res[v][g] = (sum, count), rows[g]) — in rank order.
    A dense group domain is tiny (<= 1024 slots): the combine is an all_gather, not a shuffle."""
    res0, rows0 = parts[0]
    res = [[(s, c) for (s, c) in row] for row in res0]
    rows = list(rows0)
    for r, rw in parts[1:]:
        for v, row in enumerate(r):
            for g, (s, c) in enumerate(row):
                s0, c0 = res[v][g]
                t = s0 + s
                if isinstance(t, int):   # wrapping Int64 like the device fold
                    t = (t + (1 << 63)) % (1 << 64) - (1 << 63)
                res[v][g] = (t, c0 + c)
        rows = [a + b for a, b in zip(rows, rw)]
    return res, rows


def all_combine_groups(local, device=None):
    """all_gather every rank's (res, rows) of Api.group_pipeline and fold them identically on every rank.
    f64 sums travel as f64, integer sums / counts as i64 (exact)."""
    import torch
    import torch.distributed as dist
    if _single(dist):
        return combine_groups([local])
    world = dist.get_world_size()
    res, rows = local
    nv, S = len(res), len(rows)
    isf = [isinstance(res[v][0][0], float) for v in range(nv)]
    f = torch.zeros(nv * S, dtype=torch.float64)
    i = torch.zeros(2 * nv * S + S, dtype=torch.int64)
    for v in range(nv):
        for g in range(S):
            s, c = res[v][g]
            if isf[v]:
                f[v * S + g] = s
            else:
                i[v * S + g] = s
            i[nv * S + v * S + g] = c
    for g in range(S):
        i[2 * nv * S + g] = rows[g]
    if device is not None:
        f, i = f.to(device), i.to(device)
    fs = [torch.zeros_like(f) for _ in range(world)]
    is_ = [torch.zeros_like(i) for _ in range(world)]
    dist.all_gather(fs, f)
    dist.all_gather(is_, i)
    parts = []
    for r in range(world):
        fl, il = fs[r].tolist(), is_[r].tolist()
        rr = [[((fl[v * S + g] if isf[v] else int(il[v * S + g])), int(il[nv * S + v * S + g])) for g in range(S)] for v in range(nv)]
        parts.append((rr, [int(il[2 * nv * S + g]) for g in range(S)]))
    return combine_groups(parts)


# ---------------------------------------------------------------- group-by across ranks (the one real exchange)
def group_owner(keys, world: int):
    """Owning rank of each group key: a multiplicative hash of the key bits, mod world."""
    import numpy as np
    k = np.asarray(keys).astype(np.int64).view(np.uint64)
    h = (k * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(33)
    return (h % np.uint64(world)).astype(np.int64)


def exchange_groups(keys, sums, counts, device=None):
    """all-to-all(v) of locally pre-aggregated groups held in HOST arrays (the CPU tests' gloo path; on the GPU box the
    exchange is device-resident: GroupExchange): every (key, sum, count) partial goes to rank hash(key) % world, so each
    rank ends up with ALL partials of the keys it owns.  Two collectives: the split sizes, then one packed payload of raw
    64-bit words, so every dtype travels exactly (f64 sums as their bit patterns, i64 / u64 as they are).
    Inputs / outputs: numpy arrays; the outputs have the inputs' dtypes."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if _single(dist):
        return keys, sums, counts
    world = dist.get_world_size()
    keys, sums, counts = np.ascontiguousarray(keys), np.ascontiguousarray(sums), np.ascontiguousarray(counts)
    assert keys.dtype.itemsize == 8 and sums.dtype.itemsize == 8 and counts.dtype.itemsize == 8, "64-bit columns travel as raw words"
    owner = group_owner(keys, world)
    order = np.argsort(owner, kind="stable")
    send_counts = np.bincount(owner, minlength=world).astype(np.int64)
    payload = np.stack([keys[order].view(np.int64), sums[order].view(np.int64), counts[order].view(np.int64)], axis=1)   # [n, 3] words
    t_send_counts = torch.from_numpy(send_counts)
    t_recv_counts = torch.zeros(world, dtype=torch.int64)
    if device is not None:
        t_send_counts, t_recv_counts = t_send_counts.to(device), t_recv_counts.to(device)
    dist.all_to_all_single(t_recv_counts, t_send_counts)
    recv_counts = t_recv_counts.cpu().numpy()
    t_send = torch.from_numpy(np.ascontiguousarray(payload))
    t_recv = torch.zeros((int(recv_counts.sum()), 3), dtype=torch.int64)
    if device is not None:
        t_send, t_recv = t_send.to(device), t_recv.to(device)
    dist.all_to_all_single(t_recv, t_send, output_split_sizes=[int(x) for x in recv_counts],
                           input_split_sizes=[int(x) for x in send_counts])
    r = t_recv.cpu().numpy()
    return r[:, 0].copy().view(keys.dtype), r[:, 1].copy().view(sums.dtype), r[:, 2].copy().view(counts.dtype)


def shuffle_rows_pays(local_rows: int, max_groups: int) -> bool:
    """pre-aggregating a shard only pays when it shrinks it.  With about as many groups as rows (the promise
    `max_groups` is at least half the shard's rows) the rows themselves are exchanged and aggregated once, at their owner."""
    return 2 * max_groups >= local_rows


def distributed_groupby_sum(api, keys_chunks, value_chunks, max_groups: int, device=None, shuffle_rows=None):
    """GROUP BY over row-sharded data with host-resident results: local hash aggregate -> exchange_groups -> ONE merge of
    the received partials (rdf_groupby_merge: sums added, counts added, dtypes kept) — or, when pre-aggregation cannot
    shrink the shard (`shuffle_rows`, default shuffle_rows_pays), the rows are exchanged and aggregated at their owner.
    `api` is the engine.  Returns numpy (keys, sums, counts) of the groups this rank owns, sorted by key; the union over
    ranks is the full result."""
    import numpy as np
    from ._abi import HostArray
    local_rows = sum(ch.length for ch in keys_chunks)
    if shuffle_rows is None:
        shuffle_rows = shuffle_rows_pays(local_rows, max_groups)
    if shuffle_rows and all(ch.validity is None for ch in list(keys_chunks) + list(value_chunks)):
        kk = np.concatenate([ch.to_numpy() for ch in keys_chunks]) if keys_chunks else np.zeros(0, np.int64)
        vv = np.concatenate([ch.to_numpy() for ch in value_chunks]) if value_chunks else np.zeros(0)
        wide = np.uint64 if kk.dtype.kind == "u" else np.int64
        vwide = np.float64 if vv.dtype.kind == "f" else (np.uint64 if vv.dtype.kind == "u" else np.int64)
        rk, rv, _ = exchange_groups(kk.astype(wide), vv.astype(vwide), np.ones(len(kk), dtype=np.int64), device)
        if len(rk) == 0:
            return rk, rv, np.zeros(0, dtype=np.int64)
        mk, ms, mc = api.groupby_sum([HostArray.from_numpy(rk)], [HostArray.from_numpy(rv)], max_groups)
        o = np.argsort(mk.to_numpy())
        return mk.to_numpy()[o], ms.to_numpy()[o], mc.to_numpy()[o]
    k, s, c = api.groupby_sum(keys_chunks, value_chunks, max_groups)
    if k.null_count:
        raise ValueError("distributed group-by: NULL keys are not supported in the exchange")
    kk = k.to_numpy()
    wide = np.uint64 if kk.dtype.kind == "u" else np.int64
    rk, rs, rc = exchange_groups(kk.astype(wide), s.to_numpy(), c.to_numpy(), device)
    if len(rk) == 0:
        return rk, rs, rc
    mk, ms, mc = api.groupby_merge(HostArray.from_numpy(rk), HostArray.from_numpy(rs), HostArray.from_numpy(rc), "sum", max_groups)
    o = np.argsort(mk.to_numpy())
    return mk.to_numpy()[o], ms.to_numpy()[o], mc.to_numpy()[o]


class GroupExchange:
    """The device-resident exchange of the multi-GPU GROUP BY (what stands where the reference panics,
    src/evaluation.rs:73): this rank's partial groups are bucketed by owning rank ON THE DEVICE (rdf_group_exchange_pack),
    travel with one all_to_all_single over RCCL / xGMI (device tensors in, device tensors out; only the `world` split
    sizes visit the host), and are merged where they land by one rdf_groupby_merge on device buffers.  Nothing of the
    payload is copied to the host inside a step.  With a CPU-only backend (gloo smoke runs) the payload is staged through
    host tensors for the collective alone."""

    def __init__(self, api, lib, torch, dev, comm_dev, cap: int):
        self.api, self.lib, self.torch, self.dev, self.comm_dev, self.cap = api, lib, torch, dev, comm_dev, cap
        self.packed = torch.empty(cap * 3 + 8, dtype=torch.int64, device=dev)
        self._bufs = [torch.empty(cap + 8, dtype=torch.int64, device=dev) for _ in range(3)]
        # what the last exchange moved, for the bench line: bytes this rank sent (all owners / other ranks only), bytes it
        # received, and the host-clock time from "pack" to "received and unpacked" (collectives + pack / unpack kernels)
        self.stats = {}

    def _note(self, t0, send_counts, recv_counts, width):
        import time
        import torch.distributed as dist
        me = dist.get_rank()
        self.stats = {"exchange_ms": (time.perf_counter() - t0) * 1e3,
                      "exchange_bytes_sent": int(sum(send_counts)) * width,
                      "exchange_bytes_sent_remote": int(sum(c for r, c in enumerate(send_counts) if r != me)) * width,
                      "exchange_bytes_received": int(sum(recv_counts)) * width}

    # One all_to_all_single call moves at most this many bytes out of a rank.  Measured on this image (torch 2.10 + RCCL 2.26.6,
    # tools/rccl_a2a_probe.py): a call whose buffer passes ~1 GiB delivers only the first half of it on a 1-rank communicator
    # (0.96 GB arrives whole, 1.12 GB does not) without any error — the exchange is therefore cut into rounds well below that.
    MAX_BYTES_PER_CALL = int(__import__("os").environ.get("RDF_A2A_MAX_BYTES", 256 << 20))   # (the tests shrink it to run the rounds on small inputs)

    def _all_to_all_rows(self, send, send_counts, recv_counts, width):
        """send: [n, width] int64 rows grouped by destination rank (send_counts[r] rows for rank r) -> [m, width] rows received,
        grouped by round and source rank (the consumers aggregate the rows: their order is free).  Collective."""
        import torch.distributed as dist
        torch = self.torch
        world = dist.get_world_size()
        cd = self.comm_dev if self.comm_dev is not None else "cpu"
        m = sum(recv_counts)
        chunk = max(1, self.MAX_BYTES_PER_CALL // (8 * width) // world)          # rows per (round, destination)
        # The number of rounds must be the SAME on every rank: the largest pair count s_ij is known to ranks i and j only, so a
        # rank deriving it from its own counts could take the one-call path while its peers loop (a hang, or truncated buffers).
        # One MAX all_reduce of the local maximum settles it (the native path reads it off the all-gathered matrix instead).
        t_big = torch.tensor([max(list(send_counts) + list(recv_counts) + [0])], dtype=torch.int64, device=cd)
        dist.all_reduce(t_big, op=dist.ReduceOp.MAX)
        rounds = max(1, (int(t_big.item()) + chunk - 1) // chunk)
        if rounds == 1:
            if self.comm_dev is None:
                send = send.cpu()
            recv = torch.empty((m, width), dtype=torch.int64, device=cd)
            dist.all_to_all_single(recv, send, output_split_sizes=list(recv_counts), input_split_sizes=list(send_counts))
            return recv if self.comm_dev is not None else recv.to(self.dev)
        starts = [0]
        for c in send_counts:
            starts.append(starts[-1] + c)
        recv = torch.empty((m, width), dtype=torch.int64, device=self.dev)
        at = 0
        for r in range(rounds):
            ins = [max(0, min(chunk, c - r * chunk)) for c in send_counts]
            outs = [max(0, min(chunk, c - r * chunk)) for c in recv_counts]
            part = torch.cat([send[starts[o] + r * chunk: starts[o] + r * chunk + ins[o]] for o in range(world)]) if sum(ins) else send[:0]
            if self.comm_dev is None:
                part = part.cpu()
            got = torch.empty((sum(outs), width), dtype=torch.int64, device=cd)
            dist.all_to_all_single(got, part, output_split_sizes=outs, input_split_sizes=ins)
            recv[at:at + sum(outs)] = got if self.comm_dev is not None else got.to(self.dev)
            at += sum(outs)
        return recv

    def _merged(self, kdt, sdt):
        from ._abi import I64, DeviceArray
        return tuple(DeviceArray(b.data_ptr(), None, 0, self.cap, dt, 0, keep=b) for b, dt in zip(self._bufs, (kdt, sdt, I64)))

    def exchange_and_merge(self, gk, gs, gc, max_groups: int, agg: str = "sum"):
        import torch.distributed as dist
        from ._abi import DeviceArray
        torch = self.torch
        world = dist.get_world_size()
        import time
        n = gk.length
        if 3 * n + 8 > self.packed.numel():
            # more partial groups than the exchange was sized for: grow the send buffer.  (Raising here, on ONE rank and before
            # the collectives, would leave the other ranks waiting in all_to_all_single for ever.)
            self.packed = torch.empty(3 * (n + n // 4) + 8, dtype=torch.int64, device=self.dev)
        t0 = time.perf_counter()
        K = DeviceArray(gk.values_ptr, None, 0, n, gk.dtype, 0)
        S = DeviceArray(gs.values_ptr, None, 0, n, gs.dtype, 0)
        Cn = DeviceArray(gc.values_ptr, None, 0, n, gc.dtype, 0)
        torch.cuda.current_stream().synchronize()                     # the send buffer may still be read by the previous collective
        send_counts = self.api.group_exchange_pack(K, S, Cn, world, self.packed.data_ptr())   # returns with its stream drained
        cd = self.comm_dev if self.comm_dev is not None else "cpu"
        t_sc = torch.tensor(send_counts, dtype=torch.int64, device=cd)
        t_rc = torch.zeros(world, dtype=torch.int64, device=cd)
        dist.all_to_all_single(t_rc, t_sc)
        recv_counts = [int(x) for x in t_rc.tolist()]
        m = sum(recv_counts)
        recv = self._all_to_all_rows(self.packed[:3 * n].view(n, 3), send_counts, recv_counts, 3)   # CPU-only backend: staged for the collective alone
        torch.cuda.current_stream().synchronize()                     # the library reads `recv` on its own stream
        cols = [torch.empty(m + 8, dtype=torch.int64, device=self.dev) for _ in range(3)]
        rk, rs, rc = (DeviceArray(t.data_ptr(), None, 0, m, dt, 0, keep=t, capacity=m) for t, dt in zip(cols, (gk.dtype, gs.dtype, gc.dtype)))
        torch.cuda.current_stream().synchronize()
        self.api.group_exchange_unpack(recv.data_ptr(), m, rk, rs, rc)
        self._note(t0, send_counts, recv_counts, 24)
        return self.api.groupby_merge(rk, rs, rc, agg, max_groups, outs=self._merged(gk.dtype, gs.dtype))

    def shuffle_rows_and_aggregate(self, K, V, max_groups: int, agg: str = "sum"):
        """The row-shuffle fallback: the shard's ROWS (one device chunk of 8-byte keys, one of 8-byte values, no
        NULLs) are bucketed by owner on the device (rdf_row_exchange_pack, 16 bytes per row), travel with one
        all_to_all_single, and the owner runs one rdf_groupby_agg over what it received."""
        import torch.distributed as dist
        from ._abi import DeviceArray
        torch = self.torch
        world = dist.get_world_size()
        import time
        n = K.length
        t0 = time.perf_counter()
        if getattr(self, "_rows_packed", None) is None or self._rows_packed.numel() < 2 * n + 8:
            self._rows_packed = torch.empty(2 * n + 8, dtype=torch.int64, device=self.dev)
        torch.cuda.current_stream().synchronize()
        send_counts = self.api.row_exchange_pack(K, V, world, self._rows_packed.data_ptr())
        cd = self.comm_dev if self.comm_dev is not None else "cpu"
        t_sc = torch.tensor(send_counts, dtype=torch.int64, device=cd)
        t_rc = torch.zeros(world, dtype=torch.int64, device=cd)
        dist.all_to_all_single(t_rc, t_sc)
        recv_counts = [int(x) for x in t_rc.tolist()]
        m = sum(recv_counts)
        recv = self._all_to_all_rows(self._rows_packed[:2 * n].view(n, 2), send_counts, recv_counts, 2)
        cols = [torch.empty(m + 8, dtype=torch.int64, device=self.dev) for _ in range(2)]
        rk, rv = (DeviceArray(t.data_ptr(), None, 0, m, dt, 0, keep=t, capacity=m) for t, dt in zip(cols, (K.dtype, V.dtype)))
        torch.cuda.current_stream().synchronize()
        self.api.row_exchange_unpack(recv.data_ptr(), m, rk, rv)
        self._note(t0, send_counts, recv_counts, 16)
        ko, so, co = self._merged(K.dtype, self.api._agg_out_dtype(self.api.AGGS[agg], V.dtype))
        return self.api.groupby_agg([[rk]], [rv], agg, max_groups, ([ko], so, co))
