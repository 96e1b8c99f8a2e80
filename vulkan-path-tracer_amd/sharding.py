"""Multi-GPU partition of the image and the single collective of the path.

Rows are dealt round-robin: row y belongs to rank y % world (the reference's interleaved split-screen
scheme, RayGen.slang:16-25, along one axis).  Seeds depend on (pixel, frame) only, so the assembled image
is bit-identical for any world size.  The only data-path collective is one gather of the finished shards.  In the
product it is issued by the library (include/vpt.h vpt_comm_gather_shards: ncclGather over xGMI on the context's own
stream; vpt_multi_gather_shards when one process drives all devices); ShardComm below wires it to a process-per-GPU
launch, where torch.distributed is only the control plane that carries the 128-byte communicator id.  The torch
functions further down restate the same partition for the CPU tests (gloo, world_size 2)."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _abi


class ShardComm:
    """One rank's end of the shard gather.  host_staged=False (the product path): ncclCommInitRank inside
    libvpt_hip.so, ncclGather on the render stream, row re-interleave on the root.  host_staged=True is a test hook for
    boxes with fewer devices than ranks (RCCL refuses two ranks on one device): the shards travel through host memory
    over the torch process group instead and the root assembles them with the same library call."""

    def __init__(self, pt, rank, world, root=0, host_staged=False, group=None):
        self.pt, self.rank, self.world, self.root, self.host_staged, self.group = pt, rank, world, root, host_staged, group
        self.ready = False
        if world > 1 and not host_staged:
            # two ranks on one device: refuse here, with a message, before RCCL's bootstrap is entered (it rejects the configuration
            # itself — "Duplicate GPU detected" — but only after every rank has joined)
            me = self.device_identity()
            names = [None] * world
            dist.all_gather_object(names, me, group=group)
            if len(set(names)) != world:
                raise RuntimeError("ShardComm: ranks share a device %r — one process per GPU is required (VPT_ERR_DEVICE)" % (names,))
            ident = torch.zeros(_abi.COMM_ID_BYTES, dtype=torch.uint8)
            if rank == root:
                buf = (C.c_ubyte * _abi.COMM_ID_BYTES)()
                rc = pt.lib.vpt_comm_unique_id(buf)
                if rc != 0:
                    raise RuntimeError("vpt_comm_unique_id failed: %d" % rc)
                ident = torch.tensor(list(buf), dtype=torch.uint8)
            dist.broadcast(ident, src=root, group=group)     # control plane: 128 bytes, once
            raw = bytes(ident.tolist())
            rc = pt.lib.vpt_comm_init(pt.ctx, raw, rank, world)
            if rc != 0:
                raise RuntimeError("vpt_comm_init failed: %s" % pt.lib.vpt_last_error(pt.ctx).decode())
            self.ready = True

    def device_identity(self):
        """(host name, PCI bus id) of this rank's device (vpt_device_identity)."""
        import socket
        buf = C.create_string_buffer(64)
        rc = self.pt.lib.vpt_device_identity(self.pt.ctx, buf, 64)
        if rc != 0:
            raise RuntimeError("vpt_device_identity failed: %s" % self.pt.lib.vpt_last_error(self.pt.ctx).decode())
        return (socket.gethostname(), buf.value.decode())

    def info(self):
        """vpt_comm_get_info as a dict: the mapped RCCL's version and file, the compiled-against version, nranks / rank / device as
        RCCL reports them for this communicator."""
        ci = _abi.CommInfo()
        rc = self.pt.lib.vpt_comm_get_info(self.pt.ctx, C.byref(ci))
        if rc != 0:
            raise RuntimeError("vpt_comm_get_info failed: %s" % self.pt.lib.vpt_last_error(self.pt.ctx).decode())
        return {"rccl_version_runtime": ci.rccl_version_runtime, "rccl_version_compiled": ci.rccl_version_compiled, "nranks": ci.nranks, "rank": ci.rank,
                "device": ci.device, "library_path": ci.library_path.decode()}

    def gather_and_assemble(self):
        """After this call the root context holds the whole image (radiance() / postprocess() work there)."""
        pt = self.pt
        if self.world == 1:
            return
        if not self.host_staged:
            rc = pt.lib.vpt_comm_gather_shards(pt.ctx, self.root)
            if rc != 0:
                raise RuntimeError("vpt_comm_gather_shards failed: %s" % pt.lib.vpt_last_error(pt.ctx).decode())
            return
        shard = torch.empty(pt.shard_floats(), dtype=torch.float32, device="cuda")
        pt.shard_to_device(shard.data_ptr())
        gathered = gather_shards(shard.cpu(), self.world, group=self.group)
        if self.rank == self.root:
            g = gathered.to("cuda")
            pt.assemble_shards(g.data_ptr(), self.world)

    def close(self):
        if self.ready:
            self.pt.lib.vpt_comm_destroy(self.pt.ctx)
            self.ready = False


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
