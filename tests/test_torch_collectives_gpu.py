"""parallel.TorchCollectives (the fy_collectives bench.py installs) on the one GPU of the test box:

* backend "nccl" (= RCCL) with a process group of ONE rank: the library's stream handed to torch as an ExternalStream, the
  collectives enqueued on it without host synchronisation -- the code path of the multi-GPU run, degenerate in size;
* bench.py end to end with two ranks sharing the GPU over gloo (FY_BENCH_REHEARSAL=1; RCCL refuses two ranks on one
  device): rendezvous, cooperative scoring through the callbacks, max-over-ranks timing, the one JSON line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist

from test_pruned_coop_gpu import make
from util import assert_topn_matches, pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_single_rank_group_drives_the_cooperative_path(monkeypatch):
    monkeypatch.setenv("FY_PRUNE_MIN_ITEMS", "256")
    monkeypatch.setenv("FY_M24_MIN_ITEMS", "0")
    monkeypatch.setenv("FY_SEED_CHUNKS", "1")
    monkeypatch.setenv("FY_COOP_FORCE", "1")
    P = pkg()
    par = __import__("importlib").import_module("filmyou-core_amd.parallel")
    data, clustering, conf, ref = make("ml100k", 1, "0.1", 20)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        comm = par.TorchCollectives(0)
        assert comm.nccl
        ctx = P.Context(0)
        rec = P.RM2Job(conf, ctx).run(data, clustering=clustering, rank=0, world=1, collectives=comm)
        assert comm.calls["reduce_scatter_f32"] >= 2 and comm.calls["all_gather"] >= 1, comm.calls
        assert_topn_matches(rec.rows(), ref, 20)
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_compiled_rccl_transport_single_rank(monkeypatch):
    """fy_rccl_* (csrc/fy_rccl.hip): the library's own ncclAllGather / ncclReduceScatter callbacks on the job's stream, a
    communicator of ONE rank (the test box has one GPU; RCCL refuses two ranks on one device) driving the cooperative path."""
    monkeypatch.setenv("FY_PRUNE_MIN_ITEMS", "256")
    monkeypatch.setenv("FY_M24_MIN_ITEMS", "0")
    monkeypatch.setenv("FY_SEED_CHUNKS", "1")
    monkeypatch.setenv("FY_COOP_FORCE", "1")
    P = pkg()
    par = __import__("importlib").import_module("filmyou-core_amd.parallel")
    data, clustering, conf, ref = make("ml100k", 1, "0.1", 20)
    ctx = P.Context(0)
    comm = par.RcclCollectives(ctx, 0, 1)
    rec = P.RM2Job(conf, ctx).run(data, clustering=clustering, rank=0, world=1, collectives=comm)
    calls = comm.calls
    assert calls["reduce_scatter_f32"] >= 2 and calls["all_gather"] >= 1 and calls["bytes"] > 0, calls
    assert_topn_matches(rec.rows(), ref, 20)
    rec.close()
    comm.close()
    ctx.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("forced,self_launch", [(True, False), (False, True)])
def test_bench_two_ranks_rehearsal(forced, self_launch, tmp_path):
    env = dict(os.environ, FY_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if forced:    # ML-1M shape is below the production thresholds: force the cooperative path onto it
        env.update(FY_PRUNE_MIN_ITEMS="256", FY_M24_MIN_ITEMS="0")
    # self_launch: `python bench.py --gpus 2` with WORLD_SIZE unset starts its two ranks itself (what the driver's plain
    # invocation does); otherwise the driver's torch.distributed.run command line
    launcher = [sys.executable] if self_launch else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port())]
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--shape", "ml1m", "--no-cpu"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0
    assert j["multi_gpu"]["rccl_ranks"] == 0 and "gloo" in j["multi_gpu"]["transport"]      # a rehearsal, and labelled as one
    assert j["scaling"] == "strong" and j["vs_baseline"] is None
    # both ranks' lists are in the total: 6040 users x top-50
    recs = j["value"] * j["ms_per_step"] * 1e-3
    assert 0.99 * 6040 * 50 < recs < 6040 * 50 + 1
