"""The N > 1 flow of bench.py on the one GPU a development box has: two ranks launched by torch.distributed.run share cuda:0 and talk
gloo (MLIIS_DIST_BACKEND test hook; RCCL refuses two ranks on one device).  Everything but the collective library itself is the
driver's multi-GPU path: RANK / LOCAL_RANK / WORLD_SIZE from the launcher, task t -> rank t mod P, ONE all-reduce(sum) over
[task deltas | BN contributions] per meta-step, barrier + max-over-ranks timing, rank 0 prints the one JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_run_the_sharded_meta_step_and_print_one_line():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MLIIS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline",
           "--image-size", "64"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["value"] > 0
    assert d["dist"]["world"] == 2 and d["dist"]["tasks_per_meta_step"] == 2
    assert d["dist"]["allreduce_bytes_per_meta_step"] > 8_000_000    # flat theta (2.07 M floats) + BN contributions, fp32
    assert d["cpu_baseline"] is None                                # the CPU baseline is timed at N = 1 only
    # value = all ranks' images / max-over-ranks time: 2 tasks x 64 images x steps
    assert abs(d["value"] - 2 * 64 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]


def test_plain_bench_command_with_gpus_2_launches_two_ranks_by_itself(tmp_path):
    """`python bench.py --gpus 2` -- no launcher, no WORLD_SIZE -- starts torch.distributed.run as a child, the
    line says n_gpus 2 / world 2, and `--cpu-baseline-from` carries the N = 1 line's cpu_baseline by value."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MLIIS_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = tmp_path / "BENCH_one.json"
    one.write_text(json.dumps({"parsed": {"n_gpus": 1, "cpu_baseline": {"value": 18.5, "unit": "images/s", "cores": 16, "kind": "port", "sample": "x"}}}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline", "--image-size", "64",
           "--cpu-baseline-from", str(one)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dist"]["world"] == 2 and d["dist"]["tasks_per_meta_step"] == 2 and d["value"] > 0
    assert d["cpu_baseline"]["value"] == 18.5 and "BENCH_one.json" in d["cpu_baseline"]["carried_from"]


def test_rccl_single_rank_process_group_runs_the_collectives():
    """RCCL itself on the box (world size 1 -- all a one-GPU machine allows): the process group comes up on backend "nccl", and the three
    collectives the meta-learner issues (all-reduce SUM of the comm buffer on the learner's stream, all-reduce MAX of the stop flag,
    broadcast of evaluation results) execute on device tensors."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mliis_amd.reptile import Dist
D = Dist()
assert D.world == 1 and D.rank == 0 and dist.get_backend() == "nccl"
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    t = torch.arange(2_089_228, dtype=torch.float32, device="cuda")      # the size of [theta | BN contributions]
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
s.synchronize()
assert float(t[-1]) == 2_089_227.0
f = torch.tensor([1], dtype=torch.int32, device="cuda")
dist.all_reduce(f, op=dist.ReduceOp.MAX)
b = torch.tensor([0.25, 0.5], dtype=torch.float64, device="cuda")
dist.broadcast(b, src=0)
dist.barrier()
torch.cuda.synchronize()
assert int(f.item()) == 1 and b.tolist() == [0.25, 0.5]
print("rccl ok", torch.cuda.nccl.version())
dist.destroy_process_group()
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "rccl ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
