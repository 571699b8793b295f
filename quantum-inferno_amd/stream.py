"""Long records as overlapped power-of-two chunks (BASELINE config 5: 24 h of 800 Hz infrasound = 69 120 000
samples per channel, chunks of 2^20 with a hop of 2^19).  A TFR of a chunk depends on that chunk only, so chunks are
independent work items like channels: no exchange between them, and a run can restart at any chunk boundary from
the chunk index alone (the only state a streaming run carries)."""
from typing import Iterator, Tuple

import numpy as np
import torch

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


def stream_reduced(plan: engine.TfrPlan, sig, hop: int, which: str = "cwt", first_chunk: int = 0, power_scale=1.0):
    """Run `plan.cwt` / `plan.stx` over every chunk of a long [channels, n_total] record and keep only the reduced
    product per chunk (per-band power, max / total / entropy sums): yields (chunk index, start, TfrResult)."""
    fn = {"cwt": plan.cwt, "stx": plan.stx, "cwt_atoms": plan.cwt_atoms}[which]
    scratch = None
    for i, start, view in iter_chunks(sig, plan.n, hop, first_chunk):
        x = view.to(device=plan.device, dtype=plan.rdtype, non_blocking=True).contiguous()
        res = fn(x, coef=True, reductions=True, power_scale=power_scale, out=scratch)
        scratch = res  # panel buffers are reused chunk after chunk
        yield i, start, res
