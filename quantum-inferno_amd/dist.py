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


def pack_reduced(results):
    """Flatten the reduced product of several TfrResult into one float64 buffer per rank:
    per result [C, B + n + 4] = power_band | power_time | stats."""
    parts = []
    for r in results:
        parts.append(torch.cat([r.power_band, r.power_time.to(torch.float64), r.stats], dim=1).reshape(-1))
    return torch.cat(parts)


def unpack_reduced(flat, n_channels, shapes):
    """Inverse of pack_reduced: shapes = [(B, n), ...] -> list of (power_band, power_time, stats)."""
    out, pos = [], 0
    for n_b, n in shapes:
        width = n_b + n + 4
        block = flat[pos : pos + n_channels * width].reshape(n_channels, width)
        out.append((block[:, :n_b], block[:, n_b : n_b + n], block[:, n_b + n :]))
        pos += n_channels * width
    return out


def gather_reduced(flat, dst=0, group=None):
    """Gather equal-sized reduced buffers to `dst`; returns [world, len] there, None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flat.unsqueeze(0)
    world = dist.get_world_size(group)
    if dist.get_rank(group) == dst:
        bufs = [torch.empty_like(flat) for _ in range(world)]
        dist.gather(flat, bufs, dst=dst, group=group)
        return torch.stack(bufs)
    dist.gather(flat, None, dst=dst, group=group)
    return None
