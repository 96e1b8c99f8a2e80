"""N > 1 host path on CPU: world_size-2 gloo run of the shard -> gather -> assemble logic bench.py uses."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, out_path):
    sys.path.insert(0, ROOT)
    import importlib
    sharding = importlib.import_module("vulkan-path-tracer_amd.sharding")
    vpt = importlib.import_module("vulkan-path-tracer_amd")
    from oracle import oracle_py
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every rank renders the same deterministic image with the oracle and keeps only its own rows,
    # exactly the rows a sharded HIP context would own
    sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
    o = oracle_py.Oracle(sc, w, h, threads=1)
    o.set_params(vpt.default_params(max_depth=3))
    o.render(2)
    full = torch.from_numpy(o.radiance())
    o.close()
    local = sharding.extract_rows(full, rank, world)
    assert local.numel() == sharding.shard_floats(w, h, world)
    gathered = sharding.gather_shards(local, world)
    img = sharding.assemble_rows(gathered, w, h, world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing reduction
    if rank == 0:
        np.save(out_path, np.stack([img.numpy(), full.numpy()]))
        assert t.item() == world
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("w,h", [(24, 13), (16, 8)])
def test_gloo_world2_gather_assemble(tmp_path, w, h):
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(2, _free_port(), w, h, out), nprocs=2, join=True)
    img, full = np.load(out)
    assert np.array_equal(img, full)


@pytest.mark.parametrize("world,w,h", [(4, 12, 10), (8, 8, 16), (8, 8, 19)])
def test_gloo_world4_and_8_gather_assemble(tmp_path, world, w, h):
    """The node's own shape: 4 and 8 ranks (one per GPU of an MI355X node), an even split (16 rows / 8) and ragged ones (10 / 4, 19 / 8:
    the first ranks own one row more, every shard is padded to rank 0's size for the gather)."""
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), w, h, out), nprocs=world, join=True)
    img, full = np.load(out)
    assert np.array_equal(img, full)


def test_shard_arithmetic():
    import importlib
    sys.path.insert(0, ROOT)
    sharding = importlib.import_module("vulkan-path-tracer_amd.sharding")
    for h in (1080, 2160, 37, 7):
        for world in (1, 2, 3, 4, 8):
            rows = [sharding.shard_rows(h, r, world) for r in range(world)]
            assert sum(rows) == h and max(rows) == rows[0]
            assert sharding.shard_floats(10, h, world) == rows[0] * 40
    assert [sharding.shard_rows(1080, r, 8) for r in range(8)] == [135] * 8   # 1080p splits evenly over 8 GPUs
    full = torch.arange(5 * 3 * 4, dtype=torch.float32).reshape(5, 3, 4)
    g = torch.stack([sharding.extract_rows(full, r, 2) for r in range(2)])
    assert torch.equal(sharding.assemble_rows(g, 3, 5, 2), full)


class _FakeLib:
    """Stands in for libvpt_hip.so in the control-plane test below (no device here): records what ShardComm hands to
    vpt_comm_init."""
    def __init__(self, rank):
        self.rank, self.got = rank, None

    def vpt_comm_unique_id(self, buf):
        for i in range(128):
            buf[i] = (i * 7 + 3) & 0xff
        return 0

    def vpt_comm_init(self, ctx, raw, rank, world):
        self.got = (bytes(raw), rank, world)
        return 0

    def vpt_comm_destroy(self, ctx):
        return 0

    def vpt_device_identity(self, ctx, buf, n):   # one device per rank — unless the test asks for a clash
        buf.value = (b"0000:%02x:00.0" % (5 if self.same_device else 5 + self.rank))
        return 0

    def vpt_last_error(self, ctx):
        return b""


class _FakePt:
    def __init__(self, rank, same_device=False):
        self.lib, self.ctx = _FakeLib(rank), None
        self.lib.same_device = same_device


def _id_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    sharding = importlib.import_module("vulkan-path-tracer_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pt = _FakePt(rank)
    comm = sharding.ShardComm(pt, rank, world)          # rank 0 makes the id, gloo carries it, every rank inits with it
    raw, r, w = pt.lib.got
    open(os.path.join(out_dir, "id%d.bin" % rank), "wb").write(raw + bytes([r, w]))
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


def _clash_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    sharding = importlib.import_module("vulkan-path-tracer_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pt = _FakePt(rank, same_device=True)
    try:
        sharding.ShardComm(pt, rank, world)
        msg = "no error"
    except RuntimeError as e:
        msg = str(e)
    open(os.path.join(out_dir, "clash%d.txt" % rank), "w").write(msg + "|" + repr(pt.lib.got))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_device_are_refused_before_the_communicator_is_made(tmp_path):
    """(host name, PCI bus id) of every rank travels over the control plane first: a clash raises on every rank — no rank enters
    ncclCommInitRank (where RCCL would report the duplicate only after its bootstrap), nothing hangs."""
    mp.spawn(_clash_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        msg, got = open(str(tmp_path / ("clash%d.txt" % r))).read().split("|")
        assert "share a device" in msg and "VPT_ERR_DEVICE" in msg and got == "None"


def test_comm_id_reaches_every_rank_over_the_control_plane(tmp_path):
    mp.spawn(_id_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (open(str(tmp_path / ("id%d.bin" % r)), "rb").read() for r in range(2))
    assert a[:128] == b[:128] == bytes((i * 7 + 3) & 0xff for i in range(128))
    assert (a[128], a[129]) == (0, 2) and (b[128], b[129]) == (1, 2)
