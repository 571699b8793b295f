"""One process per GPU: channels (or chunks) are independent records, so a rank transforms its own
contiguous block with no data-path collective; the only exchange is one gather of the REDUCED
product (per-band and per-time power marginals, max / total / entropy sums) to rank 0 -- the full
complex panels stay on the GPU that made them (8 x 89.7 GB at config 4 would not fit one HBM).
torch.distributed backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used for CPU tests."""
import torch
import torch.distributed as dist


def shard(total, rank, world):
    """Contiguous block [first, first+count) of `total` records owned by `rank`."""
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def reduced_slots(n_channels, n_bands, n, time_dtype=torch.float32):
    """float64 slots of one result's reduced product: power_time [C, n] in the record dtype (float32 takes half a
    slot per sample; first, so that its rows stay 16-byte aligned) | power_band [C, B] f64 | stats [C, 4] f64."""
    tbytes = n_channels * n * torch.empty((), dtype=time_dtype).element_size()
    return n_channels * n_bands + n_channels * 4 + (tbytes + 7) // 8


def pack_reduced(results):
    """The reduced product of several TfrResult as ONE float64 buffer per rank (layout: reduced_slots).  Results made
    by TfrPlan already own such a buffer (`reduced`: their power_band / stats / power_time are views into it), so this
    is at most one concatenation."""
    parts = []
    for r in results:
        blob = getattr(r, "reduced", None)
        if blob is None:
            t = r.power_time.contiguous().reshape(-1)
            pad = ((-t.numel() * t.element_size()) % 8) // t.element_size()
            if pad:
                t = torch.cat([t, t.new_zeros(pad)])
            blob = torch.cat([t.view(torch.float64), r.power_band.reshape(-1), r.stats.reshape(-1)])
        parts.append(blob)
    if len(parts) == 1:
        return parts[0]
    # results whose buffers were carved out of one allocation, in order (TfrPlan ..., reduced_out=): no copy at all
    joined = all(
        a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.storage_offset() + a.numel() == b.storage_offset()
        for a, b in zip(parts, parts[1:])
    )
    if joined:
        total = sum(p.numel() for p in parts)
        return torch.empty(0, dtype=parts[0].dtype, device=parts[0].device).set_(
            parts[0].untyped_storage(), parts[0].storage_offset(), (total,), (1,))
    return torch.cat(parts)


def unpack_reduced(flat, n_channels, shapes, time_dtype=torch.float32):
    """Inverse of pack_reduced: shapes = [(B, n), ...] -> list of (power_band, power_time, stats) views."""
    out, pos = [], 0
    for n_b, n in shapes:
        slots = reduced_slots(n_channels, n_b, n, time_dtype)
        block = flat[pos : pos + slots]
        tslots = slots - n_channels * (n_b + 4)
        time = block[:tslots].view(time_dtype)[: n_channels * n].reshape(n_channels, n)
        band = block[tslots : tslots + n_channels * n_b].reshape(n_channels, n_b)
        stats = block[tslots + n_channels * n_b :].reshape(n_channels, 4)
        out.append((band, time, stats))
        pos += slots
    return out


_GATHER_BUFFERS = {}


def gather_reduced(flat, dst=0, group=None, async_op=False, slot=0, force_collective=False):
    """Gather equal-sized reduced buffers to `dst`; returns [world, len] there, None elsewhere.  The receive buffer is
    kept between calls (one [world, len] allocation per shape and `slot`), the ranks' messages land in its rows directly.
    async_op: returns (buffer or None, work) without waiting -- `work.wait()` before `flat` is written again or the
    buffer is read (see GatherPipeline).  A group of ONE rank returns the message itself, without a collective, unless
    `force_collective` asks for the real gather-to-self (how the RCCL branch is exercised on a one-GPU box)."""
    ready = dist.is_available() and dist.is_initialized()
    if not ready or (dist.get_world_size(group) == 1 and not force_collective):
        return (flat.unsqueeze(0), None) if async_op else flat.unsqueeze(0)
    world = dist.get_world_size(group)
    out, rows = None, None
    if dist.get_rank(group) == dst:
        key = (flat.numel(), flat.dtype, flat.device, world, slot)
        out = _GATHER_BUFFERS.get(key)
        if out is None:
            for k in [k for k in _GATHER_BUFFERS if k[:4] != key[:4]]:
                del _GATHER_BUFFERS[k]
            out = _GATHER_BUFFERS[key] = torch.empty((world, flat.numel()), dtype=flat.dtype, device=flat.device)
        rows = list(out.unbind(0))
    work = dist.gather(flat, rows, dst=dst, group=group, async_op=async_op)
    return (out, work) if async_op else out


def clear_gather_buffers():
    """Free rank 0's receive buffers (they are kept between calls; a caller that changes shape wants the memory back)."""
    _GATHER_BUFFERS.clear()


def receive_bytes(message_slots, world, depth=2):
    """HBM that `depth` receive buffers of a `world`-rank gather take on the destination rank."""
    return depth * world * message_slots * 8


class GatherPipeline:
    """The gather of step k overlaps the transforms of step k + 1: `depth` message buffers are used in turn, and a
    buffer's gather is waited for (on the stream, not the host, with RCCL) just before the buffer is written again.

        pipe = GatherPipeline(depth=2)
        for k in range(steps):
            i = pipe.acquire()            # waits for the gather that last used buffer i
            ... write message[i] ...
            pipe.submit(i, message[i])    # asynchronous gather to rank `dst`
        gathered = pipe.drain()           # list of the receive buffers on `dst` (None elsewhere)
    """

    def __init__(self, depth=2, dst=0, group=None, timing=False, force_collective=False):
        self.depth, self.dst, self.group, self.force_collective = depth, dst, group, force_collective
        self.works = [None] * depth
        self.outs = [None] * depth
        self.k = 0
        # timing: how long the compute stream stood still in `acquire` waiting for a gather (an event pair around the
        # stream-side wait on a GPU, the host clock with gloo) -- what a scaling run needs to attribute a shortfall
        self.timing = timing
        self._pairs, self._host_wait = [], 0.0

    def acquire(self):
        i = self.k % self.depth
        if self.works[i] is not None:
            self._wait(self.works[i])
            self.works[i] = None
        return i

    def _wait(self, work):
        if not self.timing:
            work.wait()
            return
        if torch.cuda.is_available() and dist.get_backend(self.group) == "nccl":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            work.wait()  # (a wait of the current stream on the collective's stream, not of the host)
            b.record()
            self._pairs.append((a, b))
        else:
            import time

            t0 = time.perf_counter()
            work.wait()
            self._host_wait += time.perf_counter() - t0

    def reset_timing(self):
        self._pairs, self._host_wait = [], 0.0

    def wait_ms(self):
        """Total milliseconds the consumer waited for gathers since `reset_timing` (call after a device sync)."""
        return self._host_wait * 1e3 + sum(a.elapsed_time(b) for a, b in self._pairs)

    def submit(self, i, flat):
        self.outs[i], self.works[i] = gather_reduced(flat, self.dst, self.group, async_op=True, slot=i,
                                                          force_collective=self.force_collective)
        self.k += 1
        return self.outs[i]

    def drain(self):
        for i, w in enumerate(self.works):
            if w is not None:
                self._wait(w)
                self.works[i] = None
        return self.outs
