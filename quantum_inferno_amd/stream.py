"""Long records as overlapped power-of-two chunks (BASELINE config 5: 24 h of 800 Hz infrasound = 69 120 000
samples per channel, 1024 channels, chunks of 2^20 with a hop of 2^19, float64).

A TFR of a chunk depends on that chunk only, so (channel block, chunk) pairs are independent work items like channels:
no exchange between them, and a run can restart at any item from its index alone (the only state a streaming run
carries).  The records stay on the HOST (a NumPy array or memmap: 1024 x 69.12 M float64 is 566 GB, twice an HBM);
`StreamPipeline` moves one item at a time through two pinned staging buffers and two device buffers: the host-to-device
copy of item k + 1 runs on a copy stream while item k is transformed, so PCIe (8 MB per record and chunk, ~0.2 ms) hides
behind the transforms (~6 ms per record and chunk in float64).  Only the REDUCED product of a chunk is kept -- no panel
is ever stored (coef=False): per-band power, per-time power, max / total / entropy sums.  Items are dealt to ranks by
`dist.shard` (one process per GPU, no data-path collective; the reduced products are gathered as in dist.py)."""
from dataclasses import dataclass
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch

from . import dist as qdist
from . import engine


def chunk_starts(n_total: int, chunk: int, hop: int) -> np.ndarray:
    """Start sample of every chunk: 0, hop, 2 hop, ... while a whole chunk fits, plus one last chunk flush with the
    end of the record when the tail is not covered (it overlaps its predecessor by more than chunk - hop)."""
    if n_total < chunk:
        raise ValueError(f"record of {n_total} samples is shorter than one chunk of {chunk}")
    if not 0 < hop <= chunk:
        raise ValueError("hop must be in (0, chunk]")
    starts = np.arange(0, n_total - chunk + 1, hop, dtype=np.int64)
    if starts[-1] + chunk < n_total:
        starts = np.append(starts, n_total - chunk)
    return starts


def iter_chunks(sig, chunk: int, hop: int, first_chunk: int = 0) -> Iterator[Tuple[int, int, torch.Tensor]]:
    """Yield (chunk index, start sample, view [channels, chunk]) from chunk `first_chunk` on (the restart cursor)."""
    t = sig if isinstance(sig, torch.Tensor) else torch.from_numpy(np.asarray(sig))
    if t.dim() == 1:
        t = t.unsqueeze(0)
    starts = chunk_starts(t.shape[-1], chunk, hop)
    for i in range(first_chunk, len(starts)):
        s = int(starts[i])
        yield i, s, t[:, s : s + chunk]


def work_items(n_channels: int, block: int, n_total: int, chunk: int, hop: int) -> List[Tuple[int, int, int, int]]:
    """Every (first channel, channel count, chunk index, start sample) of a [n_channels, n_total] record set, channel
    blocks of `block` records, block-major (a rank's contiguous share then walks whole blocks through their chunks)."""
    starts = chunk_starts(n_total, chunk, hop)
    items = []
    for c0 in range(0, n_channels, block):
        cb = min(block, n_channels - c0)
        for i, s in enumerate(starts):
            items.append((c0, cb, i, int(s)))
    return items


def rank_items(items, rank: int, world: int):
    """The contiguous share of `items` that `rank` of `world` owns (dist.shard)."""
    first, count = qdist.shard(len(items), rank, world)
    return items[first : first + count]


def stream_reduced(plan: engine.TfrPlan, sig, hop: int, which: str = "cwt", first_chunk: int = 0, power_scale=1.0):
    """Run `plan.cwt` / `plan.stx` over every chunk of a long [channels, n_total] record and keep only the reduced
    product per chunk (per-band power, max / total / entropy sums): yields (chunk index, start, TfrResult).  The
    yielded reduced product is a copy (a consumer may keep every item); no panel is stored."""
    fn = {"cwt": plan.cwt, "stx": plan.stx, "cwt_atoms": plan.cwt_atoms}[which]
    scratch = None
    for i, start, view in iter_chunks(sig, plan.n, hop, first_chunk):
        x = view.to(device=plan.device, dtype=plan.rdtype, non_blocking=True).contiguous()
        scratch = fn(x, coef=False, reductions=True, power_scale=power_scale, out=scratch)
        yield i, start, _copy_reduced(scratch)


def _copy_reduced(res: engine.TfrResult) -> engine.TfrResult:
    """A result that owns its reduced product (the buffers of `res` are written again by the next item)."""
    out = engine.TfrResult(frequency_hz=res.frequency_hz, power_scale=res.power_scale)
    out.reduced = res.reduced.clone()
    n_ch, n_b = res.power_band.shape
    o1 = out.reduced.numel() - n_ch * (n_b + 4)
    o2 = o1 + n_ch * n_b
    n = res.power_time.shape[1]
    out.power_time = out.reduced[:o1].view(res.power_time.dtype)[: n_ch * n].view(n_ch, n)
    out.power_band = out.reduced[o1:o2].view(n_ch, n_b)
    out.stats = out.reduced[o2:].view(n_ch, 4)
    return out


@dataclass
class StreamItem:
    index: int          # position in the item list (the restart cursor: pass index + 1 as `first_item`)
    first_channel: int
    channels: int
    chunk: int
    start: int          # first sample of the chunk
    cwt: Optional[engine.TfrResult]
    stx: Optional[engine.TfrResult]


class StreamPipeline:
    """Double-buffered host -> device streaming of (channel block, chunk) items through a plan.

        plan = TfrPlan(1 << 20, torch.float64, ...); plan.set_styx_bank(12, 800.0); plan.set_stx_bands(12, 800.0)
        pipe = StreamPipeline(plan, host_records, hop=1 << 19, block=16, transforms=("cwt", "stx"))
        for item in pipe.run(rank=r, world=w):          # item.cwt.power_band [channels, bands], .entropy_bits, ...
            ...

    `host_records`: NumPy array / memmap [channels, n_total] (any real dtype; converted to the plan's on the way into
    the pinned buffer).  Every yielded item owns its reduced products.  keep_time=False: the per-time power of a chunk
    (8 MB per record in float64) is neither kept nor computed (`reductions="band"`: band powers, maximum, total, entropy)."""

    def __init__(self, plan: engine.TfrPlan, host_records, hop: int, block: int = 16, transforms=("cwt", "stx"),
                 power_scale: float = 1.0, keep_time: bool = True):
        self.plan, self.hop, self.block = plan, int(hop), int(block)
        # an ndarray / memmap, or any object with a 2-D `shape` that answers [channel slice, sample slice] with an array
        # (a rank of the 24 h job materialises only the records it owns)
        self.sig = host_records if hasattr(host_records, "shape") and hasattr(host_records, "__getitem__") else np.asarray(host_records)
        if len(self.sig.shape) == 1:
            self.sig = self.sig[None, :]
        self.transforms = tuple(transforms)
        for t in self.transforms:
            if t not in ("cwt", "stx"):
                raise ValueError(f"unknown transform {t!r}: 'cwt' and / or 'stx'")
        self.power_scale, self.keep_time = power_scale, keep_time
        # (no per-time power kept: it is not computed either -- no per-time planes written or summed, round 5)
        self._reductions = True if keep_time else "band"
        self.items = work_items(self.sig.shape[0], self.block, self.sig.shape[1], plan.n, self.hop)
        np_dtype = np.float64 if plan.rdtype == torch.float64 else np.float32
        # (a plan on the CPU exists only as the stand-in of bench.py --stub: the item walk without streams or pinning)
        self._gpu = torch.device(plan.device).type == "cuda"
        self._pinned = [torch.empty((self.block, plan.n), dtype=plan.rdtype) for _ in range(2)]
        if self._gpu:
            self._pinned = [p.pin_memory() for p in self._pinned]
        self._pinned_np = [p.numpy() for p in self._pinned]
        assert self._pinned_np[0].dtype == np_dtype
        self._dev = [torch.empty((self.block, plan.n), dtype=plan.rdtype, device=plan.device) for _ in range(2)]
        self._copy_stream = torch.cuda.Stream(device=plan.device) if self._gpu else None
        self._copied = [torch.cuda.Event() for _ in range(2)] if self._gpu else None    # H2D of buffer j done
        self._consumed = [torch.cuda.Event() for _ in range(2)] if self._gpu else None  # transforms that read buffer j done
        self._used = [False, False]
        self._out = {}  # (transform, channels) -> result buffers reused item after item

    def _stage(self, j, item):
        """Host gather of one item into pinned buffer j and its asynchronous copy to device buffer j."""
        c0, cb, _, s = item
        if not self._gpu:
            np.copyto(self._pinned_np[j][:cb], self.sig[c0 : c0 + cb, s : s + self.plan.n], casting="same_kind")
            self._dev[j][:cb].copy_(self._pinned[j][:cb])
            return
        if self._used[j]:
            self._copied[j].synchronize()  # the previous copy out of this pinned buffer has finished
        np.copyto(self._pinned_np[j][:cb], self.sig[c0 : c0 + cb, s : s + self.plan.n], casting="same_kind")
        with torch.cuda.stream(self._copy_stream):
            if self._used[j]:
                self._copy_stream.wait_event(self._consumed[j])  # the transforms that read device buffer j are done
            self._dev[j][:cb].copy_(self._pinned[j][:cb], non_blocking=True)
            self._copied[j].record(self._copy_stream)
        self._used[j] = True

    def run(self, rank: int = 0, world: int = 1, first_item: int = 0) -> Iterator[StreamItem]:
        mine = rank_items(self.items, rank, world)[first_item:]
        if not mine:
            return
        compute = torch.cuda.current_stream(self.plan.device) if self._gpu else None
        self._stage(0, mine[0])
        for k, item in enumerate(mine):
            j = k & 1
            c0, cb, chunk, s = item
            if self._gpu:
                compute.wait_event(self._copied[j])
            x = self._dev[j][:cb]
            res = {}
            if self.transforms == ("cwt", "stx"):
                key = ("both", cb)
                self._out[key] = self.plan.cwt_stx(x, coef=False, reductions=self._reductions, power_scale=self.power_scale,
                                                   out=self._out.get(key))
                res["cwt"], res["stx"] = self._out[key]
            else:
                for t in self.transforms:
                    key = (t, cb)
                    fn = self.plan.cwt if t == "cwt" else self.plan.stx
                    self._out[key] = fn(x, coef=False, reductions=self._reductions, power_scale=self.power_scale, out=self._out.get(key))
                    res[t] = self._out[key]
            if self._gpu:
                self._consumed[j].record(compute)
            # the next item's host gather (a 134 MB memcpy at 16 float64 records: milliseconds of host time) and its copy to
            # the device run AFTER this item's launches are queued: they overlap its transforms even when the consumer
            # waits for every item before asking for the next (round 3 staged first: the GPU idled through the gather)
            if k + 1 < len(mine):
                self._stage(j ^ 1, mine[k + 1])
            kept = {t: self._keep(r) for t, r in res.items()}
            yield StreamItem(first_item + k, c0, cb, chunk, s, kept.get("cwt"), kept.get("stx"))

    def _keep(self, res):
        if self.keep_time:
            return _copy_reduced(res)
        small = engine.TfrResult(frequency_hz=res.frequency_hz, power_scale=res.power_scale)
        small.power_band, small.stats = res.power_band.clone(), res.stats.clone()
        return small
