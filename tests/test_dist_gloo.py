"""world_size-2 (and 3, ragged) CPU rehearsal of the multi-GPU layout: contiguous batch shards
per rank + one all-gather of the descriptor matrix (gloo here, RCCL on the GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_total, width, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from imgan.shard import shard_range, gather_descriptors
    lo, hi = shard_range(n_total, rank, world)
    # the "descriptor" of image i is a deterministic function of i alone
    rows = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1) * 10.0 + \
        torch.arange(width, dtype=torch.float32).view(1, -1)
    full = gather_descriptors(rows, n_total=n_total)
    full2 = gather_descriptors(rows)                      # sizes discovered by a collective
    from imgan.shard import gather_descriptors_async
    pending = gather_descriptors_async(rows, n_total=n_total)
    rows.fill_(-1.0)                                      # the caller may reuse its buffer at once
    assert torch.equal(pending.result(), full)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    assert torch.equal(full, full2)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 8), (2, 7), (3, 10)])
def test_sharded_gather_reassembles_image_order(tmp_path, world, n_total):
    width = 5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, width, str(tmp_path)), nprocs=world, join=True)
    want = np.arange(n_total, dtype=np.float32)[:, None] * 10.0 + np.arange(width, dtype=np.float32)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got, want)                  # every rank holds the full index


def test_shard_ranges_partition_the_batch():
    sys.path.insert(0, os.path.join(ROOT, "image-analysis_amd"))
    from imgan.shard import shard_range
    for n in (0, 1, 7, 8, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)
