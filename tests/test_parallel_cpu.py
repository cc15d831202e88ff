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


# ---------------------------------------------------------------- the cooperative decomposition (fy_collectives)
def _coop_worker(rank, world, port, golden_path, out_dir):
    """Every rank holds a ROW RANGE of M and evaluates every user's partial log-sums over the rated items in its rows;
    reduce_scatter_f32 (parallel.TorchCollectives, gloo, host pointers) hands each user's owner the complete sums.
    numpy stands in for the HIP kernels; the collective contract and the decomposition are what is tested."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    par = importlib.import_module("filmyou-core_amd.parallel")
    par.init_distributed(backend="gloo")
    import json
    g = json.load(open(golden_path))
    A = np.asarray(g["A_items_by_users"], dtype=np.float64).T              # users x items, one cluster (all users)
    U, I = A.shape
    lam, n_items_conf = 0.1, I
    su = A.sum(1)
    X = A / su[:, None]
    p = A.sum(0) / np.floor(su).sum()
    b = X.sum(0)
    M = (1 - lam) ** 2 * (X.T @ X) + lam * (1 - lam) * np.outer(p, b)       # M[j][i]
    r0, r1 = rank * I // world, (rank + 1) * I // world                    # my item rows
    bounds = [k * U // world for k in range(world + 1)]                    # user ownership
    umax = max(bounds[k + 1] - bounds[k] for k in range(world))
    send = np.zeros((world, umax, I), dtype=np.float32)
    for k in range(world):
        for u in range(bounds[k], bounds[k + 1]):
            J = np.flatnonzero(A[u] > 0)
            mine = J[(J >= r0) & (J < r1)]
            e = (1 - lam) * (b[mine] - X[u, mine]) + lam * (U - 1) * p[mine]
            part = np.log(M[mine, :] + np.outer(e, lam * p)).sum(0)
            if rank == 0:                                                   # pvpi enters the sum once
                part = part + (len(J) - 1) * np.log(n_items_conf) - len(J) * np.log(U)
            part[mine] = np.nan                                             # the row's holder masks the rated candidate
            send[k, u - bounds[k]] = part
    recv = np.full((umax, I), -1.0, dtype=np.float32)
    coll = par.TorchCollectives(None)
    coll.reduce_scatter_f32(send.ctypes.data, recv.ctypes.data, umax * I, 0)
    # all-gather contract: rank-major byte segments
    mine8 = np.full(5, rank, dtype=np.uint8)
    got = np.zeros(5 * world, dtype=np.uint8)
    coll.all_gather(mine8.ctypes.data, got.ctypes.data, 5, 0)
    assert got.reshape(world, 5).tolist() == [[k] * 5 for k in range(world)]
    np.save(os.path.join(out_dir, "scores_%d.npy" % rank), recv[:bounds[rank + 1] - bounds[rank]])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3])
def test_cooperative_partial_sums_reduce_to_the_oracle_scores(tmp_path, world):
    import oracle
    port = _free_port()
    golden = os.path.join(ROOT, "tests", "golden", "rm_test_data.json")
    mp.spawn(_coop_worker, args=(world, port, golden, str(tmp_path)), nprocs=world, join=True)
    scores = np.concatenate([np.load(tmp_path / ("scores_%d.npy" % r)) for r in range(world)])     # users x items
    import json
    A = np.asarray(json.load(open(golden))["A_items_by_users"], dtype=np.float64).T
    U, I = A.shape
    uu, ii = np.nonzero(A)
    ref = oracle.rm2(uu + 1, ii + 1, A[uu, ii].astype(np.float32), lam=0.1, number_of_items=I,
                     number_of_recommendations=1 << 30, number_of_clusters=1)
    assert scores.shape == (U, I)
    got = scores[ref["rec_user"] - 1, ref["rec_item"] - 1].astype(np.float64)
    np.testing.assert_allclose(got, ref["rec_score"].astype(np.float64), rtol=1e-5)
    # rated candidates stayed masked through the sum, everything else is a score
    assert np.array_equal(np.isnan(scores), A > 0)


def _worker_rccl_id_failure(rank, world, port, out_dir):
    """RcclCollectives when rank 0 cannot produce the ncclUniqueId: the failure travels in the broadcast, EVERY rank raises."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    par = importlib.import_module("filmyou-core_amd.parallel")
    native = importlib.import_module("filmyou-core_amd._native")
    par.init_distributed(backend="gloo")

    class _Lib:      # the library with a unique-id call that fails (what a host without librccl sees); nothing else is reached
        def fy_rccl_unique_id(self, buf):
            return -9

        def fy_last_error(self):
            return b"librccl.so could not be opened: injected"

        def fy_rccl_create(self, *a):
            raise AssertionError("rank %d went on to ncclCommInitRank: it would wait for the others forever" % rank)

    native_load = native.load
    native.load = lambda: _Lib()
    try:
        try:
            par.RcclCollectives(None, rank, world)
            verdict = "no error"
        except RuntimeError as e:
            verdict = str(e)
    finally:
        native.load = native_load
    with open(os.path.join(out_dir, "verdict_%d.txt" % rank), "w") as f:
        f.write(verdict)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_rccl_unique_id_failure_reaches_every_rank(tmp_path):
    world = 2
    mp.spawn(_worker_rccl_id_failure, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    verdicts = [open(tmp_path / ("verdict_%d.txt" % r)).read() for r in range(world)]
    assert all("fy_rccl_unique_id" in v and "injected" in v for v in verdicts), verdicts
