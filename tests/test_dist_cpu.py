"""CPU, world_size 2 over gloo: the multi-GPU path (contiguous row shards per rank, one result
gather to rank 0) is exercised without GPUs."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from generativeimage2text_amd import inference
    import bench
    n = 11
    s, e = inference.shard_range(n, inference.get_mpi_rank(), inference.get_mpi_size())
    rows = [["key%d" % i, "cap%d" % i] for i in range(s, e)]
    allrows = inference._gather_rows(rows)
    # bench.py's token gather: every rank contributes [B, T] tokens + [B] logprobs
    toks = torch.full((3, 5), rank, dtype=torch.int64)
    lps = torch.full((3,), float(rank))
    g_t, g_l = bench.gather_results(toks, lps)
    if rank == 0:
        assert [r[0] for r in allrows] == ["key%d" % i for i in range(n)]
        assert g_t.shape == (world * 3, 5) and g_t[3:].eq(1).all() and g_t[:3].eq(0).all()
        assert g_l.tolist() == [0.0] * 3 + [1.0] * 3
        open(os.path.join(tmpdir, "ok"), "w").write("1")
    else:
        assert allrows is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
