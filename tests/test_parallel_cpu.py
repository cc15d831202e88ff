"""world_size-2 `gloo` rehearsal (CPU) of the only exchange step of the sharded RM2 job: the all-gather of per-rank
partial item statistics and their fixed-order combination (parallel.all_gather_stats / combine_in_rank_order =
what RCCL + fy_rm2_set_global_stats do on the GPUs).  The partial sums are computed here with numpy from each rank's
user shard -- test scaffolding standing in for the HIP statistics kernel, which needs a GPU."""
import importlib
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


def _worker(rank, world, port, golden_path, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    par = importlib.import_module("filmyou-core_amd.parallel")
    r, lr, w = par.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    import json
    g = json.load(open(golden_path))
    A = np.asarray(g["A_items_by_users"], dtype=np.float64)            # items x users
    n_users = A.shape[1]
    lo, hi = rank * n_users // world, (rank + 1) * n_users // world     # this rank's user range
    part = np.concatenate([A[:, lo:hi].sum(1), [np.floor(A[:, lo:hi].sum(0)).sum() * 100.0]])
    gathered = par.all_gather_stats(torch.from_numpy(part))
    assert gathered.shape == (world * len(part),)
    # rank-major layout: my own slice sits at my rank
    assert torch.equal(gathered.view(world, -1)[rank], torch.from_numpy(part))
    total = par.combine_in_rank_order(gathered, world).numpy()
    item_coll = total[:-1] / (total[-1] / 100.0)
    np.save(os.path.join(out_dir, "coll_%d.npy" % rank), item_coll)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_all_gather_of_item_statistics_world2(tmp_path, rm_golden):
    world = 2
    port = _free_port()
    golden = os.path.join(ROOT, "tests", "golden", "rm_test_data.json")
    mp.spawn(_worker, args=(world, port, golden, str(tmp_path)), nprocs=world, join=True)
    colls = [np.load(tmp_path / ("coll_%d.npy" % r)) for r in range(world)]
    np.testing.assert_array_equal(colls[0], colls[1])                         # every rank ends with the same p(i|C)
    np.testing.assert_allclose(colls[0], np.asarray(rm_golden["itemColl"]), rtol=1e-15)


def test_single_process_degenerates():
    par = importlib.import_module("filmyou-core_amd.parallel")
    t = torch.arange(5, dtype=torch.float64)
    assert torch.equal(par.all_gather_stats(t), t)
    assert torch.equal(par.combine_in_rank_order(torch.cat([t, t, t]), 3), 3 * t)
    assert par.env_world() == (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)),
                               int(os.environ.get("WORLD_SIZE", 1)))
