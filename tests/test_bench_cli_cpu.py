"""bench.py's launch contract, the part that needs no GPU: `--gpus N` is honoured or refused, never ignored."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_world_size_that_disagrees_with_gpus_exits_non_zero():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "WORLD_SIZE" in r.stderr and r.stdout.strip() == ""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],          # default --gpus 1 under a 2-rank launcher
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and r.stdout.strip() == ""


def test_check_world_decides_launch_or_run():
    b = _bench()
    assert b.check_world(b.parse(["--gpus", "1"]), {}) == "run"
    assert b.check_world(b.parse(["--gpus", "8"]), {}) == "launch"
    assert b.check_world(b.parse(["--gpus", "8"]), {"WORLD_SIZE": "8"}) == "run"
    assert b.check_world(b.parse(["--gpus", "1"]), {"WORLD_SIZE": "1"}) == "run"
    with pytest.raises(SystemExit) as e:
        b.check_world(b.parse(["--gpus", "8"]), {"WORLD_SIZE": "1"})
    assert e.value.code == 2
    with pytest.raises(SystemExit):
        b.check_world(b.parse(["--gpus", "0"]), {})


def test_launch_ranks_starts_the_launcher_as_a_child_and_relays_line_and_code(monkeypatch, capsys):
    b = _bench()
    seen = {}

    def fake_run(cmd, env=None, stdout=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=seen.get("rc", 0), stdout=b'noise\n{"n_gpus": 4, "value": 1.0}\n')
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    rc = b.launch_ranks(b.parse(argv), argv)
    out = capsys.readouterr()
    assert rc == 0 and out.out.strip() == '{"n_gpus": 4, "value": 1.0}' and "noise" in out.err
    c = seen["cmd"]
    assert c[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in c
    assert c[c.index("--nproc-per-node") + 1] == "4" and c[c.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(c[c.index("--master-port") + 1]) < 65536
    assert c[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and c[-len(argv):] == argv
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    seen["rc"] = 7
    assert b.launch_ranks(b.parse(argv), argv) == 7


def test_plain_multi_gpu_command_fails_loudly_without_gpus():
    """Here (no GPU) the self-launched ranks must die with the 'needs an MI355X' message and a non-zero code: not a silent 1-GPU line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["MLIIS_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-roofline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_bench_dist_gpu.py")
    assert r.returncode != 0 and r.stdout.strip() == "" and "needs an MI355X" in r.stderr


def test_launch_ranks_times_the_cpu_baseline_in_the_parent_and_hands_it_to_rank_0(monkeypatch, capsys):
    """An N > 1 line carries `cpu_baseline` without --cpu-baseline-from: the launching parent (which never touches the GPU) times the
    CPU oracle before it starts the ranks and passes the object by value in MLIIS_BENCH_CPU_BASELINE."""
    import json
    b = _bench()
    seen = {}

    def fake_run(cmd, env=None, stdout=None):
        seen["env"] = env
        return types.SimpleNamespace(returncode=0, stdout=b'{"n_gpus": 2}\n')
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(b, "_cpu_baseline_fn", lambda args: {"value": 7.5, "unit": "images/s", "cores": 3, "kind": "port", "sample": "stub"})
    monkeypatch.delenv("MLIIS_BENCH_CPU_BASELINE", raising=False)
    argv = ["--gpus", "2", "--steps", "1", "--warmup", "0"]
    assert b.launch_ranks(b.parse(argv), argv) == 0
    got = json.loads(seen["env"]["MLIIS_BENCH_CPU_BASELINE"])
    assert got["value"] == 7.5 and got["cores"] == 3 and "parent" in got["timed_in"]
    # --no-cpu-baseline / an explicit --cpu-baseline-from: the parent times nothing
    for extra in (["--no-cpu-baseline"], ["--cpu-baseline-from", "BENCH_r04.json"]):
        seen.clear()
        monkeypatch.setattr(b, "_cpu_baseline_fn", lambda args: (_ for _ in ()).throw(AssertionError("must not be timed")))
        assert b.launch_ranks(b.parse(argv + extra), argv + extra) == 0
        assert "MLIIS_BENCH_CPU_BASELINE" not in seen["env"]
    capsys.readouterr()


def test_cpu_baseline_follows_the_protocol_of_baseline_md():
    """at least 3 warm-up steps, then at least 2 full tasks -- checked on the sample string of a tiny run."""
    b = _bench()
    args = b.parse(["--image-size", "32", "--inner-iters", "2", "--inner-batch", "2", "--shots", "2"])
    out = b.cpu_baseline(args)
    assert out["kind"] == "port" and out["cores"] >= 1 and out["value"] > 0
    assert "after 3 warm-up steps" in out["sample"] and out["sample"].startswith("4 inner SGD steps = 2 tasks")
