"""Multi-GPU partition of the image and the single collective of the path.

Rows are dealt round-robin: row y belongs to rank y % world (the reference's interleaved split-screen
scheme, RayGen.slang:16-25, along one axis).  Seeds depend on (pixel, frame) only, so the assembled image
is bit-identical for any world size.  The only data-path collective is one gather of the finished shards
(RCCL over xGMI with backend "nccl"; gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_rows(height, rank, world):
    return (height - rank + world - 1) // world if rank < height else 0


def shard_floats(width, height, world):
    """Every rank's shard padded to rank 0's row count (== vpt_shard_floats)."""
    return shard_rows(height, 0, world) * width * 4


def gather_shards(local, world, group=None):
    """local: 1-D float32 tensor of shard_floats elements on any device -> [world, shard_floats] on every rank."""
    if world == 1:
        return local.reshape(1, -1)
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local, group=group)
    return torch.stack(out, 0)


def assemble_rows(gathered, width, height, world):
    """Reference (torch) form of vpt_assemble_shards / k_scatter_rows: [world, shard_floats] -> [H, W, 4]."""
    full = torch.empty((height, width, 4), dtype=gathered.dtype, device=gathered.device)
    for r in range(world):
        rows = shard_rows(height, r, world)
        full[r::world] = gathered[r, : rows * width * 4].reshape(rows, width, 4)
    return full


def extract_rows(full, rank, world):
    """Inverse: the padded shard buffer of `rank` from a whole [H, W, 4] image."""
    h, w = full.shape[0], full.shape[1]
    out = torch.zeros(shard_floats(w, h, world), dtype=full.dtype, device=full.device)
    part = full[rank::world].reshape(-1)
    out[: part.numel()] = part
    return out
