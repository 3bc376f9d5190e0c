"""The N>1 path on CPU: tile sharding + the all-gather of per-tile maps with world_size 2 (and 3)
over gloo. The per-tile "forward" is replaced by a deterministic function of the tile's pixels so
that the test needs no GPU; what is checked is the distributed plumbing: every rank ends with the
maps of ALL tiles, bit-identical to the serial loop's stack and in the reference's row-major order
(sw_processing.py:151-163, 235-258), including the uneven 49 = 25 + 24 and 7 = 3 + 3 + 1 splits."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vit_oracle as O
from vit_ocm_wmsegmentation_amd import sw_processing as sw


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tile_feature(slab, y0, x0, window):
    """Stand-in for the per-tile map: (2, 3) statistics of the window's pixels."""
    t = slab[:, y0:y0 + window, x0:x0 + window]
    return torch.stack([t.mean((1, 2)), t.amax((1, 2))])


def _serial(slab, window, stride):
    return torch.stack([_tile_feature(slab, y, x, window)
                        for y, x in O.sliding_window_origins(slab.shape[1], slab.shape[2], stride)])


def _worker(rank, world, port, size, window, stride, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        slab = torch.rand(3, size, size, generator=g)
        origins = sw.sliding_window_origins(size, size, stride)
        T = origins.shape[0]
        begin, end, share = sw.shard_range(T, world, rank)
        local = torch.zeros((share, 2, 3))
        for i, j in enumerate(range(begin, end)):
            local[i] = _tile_feature(slab, int(origins[j, 0]), int(origins[j, 1]), window)
        maps = sw.gather_tile_maps(local, T)
        np.save(os.path.join(out_dir, f"maps_{rank}.npy"), maps.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size", [(2, 1152), (3, 640), (2, 384)])
def test_sharded_sweep_equals_serial(tmp_path, world, size):
    window, stride = 384, 128
    port = _free_port()
    mp.spawn(_worker, args=(world, port, size, window, stride, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(7)
    ref = _serial(torch.rand(3, size, size, generator=g), window, stride).numpy()
    assert ref.shape[0] == len(range(0, size - 2 * stride, stride)) ** 2
    for r in range(world):
        got = np.load(tmp_path / f"maps_{r}.npy")
        assert got.shape == ref.shape and np.array_equal(got, ref), f"rank {r}"


def test_gather_is_identity_without_process_group():
    local = torch.arange(12.).reshape(4, 3)
    assert torch.equal(sw.gather_tile_maps(local, 3), local[:3])
