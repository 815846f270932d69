"""Meta-learner host logic on the CPU oracle learner (float64): Reptile / FOMAML outer-update algebra, sequential BN
moving-average semantics under task sharding, and the world_size-2 gloo path == single-process result."""
import copy
import os
import random
import socket

import numpy as np
import pytest
import torch

from mliis_amd import metaseg
from mliis_amd.reptile import FOMLIS, Gecko
from oracle import efficientlab_ref as R

H = 32


def _tasks(n, shots):
    out = []
    for i in range(n):
        x, y = metaseg.synthetic_task(shots, H, seed=50 + i, block=4)
        out.append(metaseg.DeviceTask("t%d" % i, torch.tensor(x), torch.tensor(y)))
    return out


def _learner():
    return R.OracleLearner(image_size=H, seed=3, dtype=torch.float64, lr=1e-2, drop_connect=False)


def _sequential_reference(L, tasks_batches, fomaml, eps, lr=None):
    """What the reference does: tasks one after another, BN moving stats never reset, numpy-style averaging."""
    old = L.export_trainable()
    ups = []
    for x, y, batches in tasks_batches:
        L.load_task(x, y)
        last = None
        for j, b in enumerate(batches):
            if fomaml and j == len(batches) - 1:
                last = L.export_trainable()
            if lr is not None:      # reptile.py:114-116 / :639-641
                L.inner_step(b, lr=lr)
                if fomaml:
                    continue
            L.inner_step(b)         # :120-121 runs also when lr was given (Gecko only: quirk E1)
        ups.append(L.export_trainable() - (last if fomaml else old))
        L.import_trainable(old)
    L.import_trainable(old + eps * torch.stack(ups).mean(0))


@pytest.mark.parametrize("fomaml,lr", [(False, None), (True, None), (False, 5e-3), (True, 5e-3)])
def test_meta_step_matches_sequential_reference(fomaml, lr):
    """lr given: Gecko runs two optimizer steps (= two BN moving-average updates) per batch, FOMLIS one; the BN recombination
    weights count updates, not batches."""
    tasks = _tasks(3, 6)
    B, iters, bs, eps = 3, 3, 4, 0.25
    A = _learner()
    ref = copy.deepcopy(A)
    if fomaml:
        meta = FOMLIS(A, train_shots=6, tail_shots=2, rng_mode="per_task", seed=1)
    else:
        meta = Gecko(A, rng_mode="per_task", seed=1)
    # replay the same per-task generators to build the reference schedule
    from mliis_amd.reptile import _task_rng
    sched = []
    for t in range(B):
        rng = _task_rng(1, 0, t)
        (x, y) = metaseg.sample_task(tasks, 6, rng)
        batches = metaseg.fomaml_batch_indices(6, 2, bs, iters, False, rng) if fomaml else \
            [list(b) for b in metaseg.mini_batch_indices(6, bs, iters, False, rng)]
        sched.append((x, y, batches))
    meta.train_step(tasks, num_shots=6, inner_batch_size=bs, inner_iters=iters, replacement=False, meta_step_size=eps, meta_batch_size=B,
                    lr=lr)
    _sequential_reference(ref, sched, fomaml, eps, lr)
    assert torch.allclose(A.export_trainable(), ref.export_trainable(), rtol=0, atol=1e-12)
    # BN moving statistics follow the SEQUENTIAL exponential average over all tasks' steps
    assert torch.allclose(A.export_bn(), ref.export_bn(), rtol=1e-10, atol=1e-12)
    assert meta.meta_iter == 1


def test_reference_rng_mode_consumes_global_random_like_the_reference(golden):
    """rng_mode='reference': task choice + batch order come from the global `random`, in the reference's order."""
    tasks = _tasks(2, 5)
    A = _learner()
    seen = []
    orig = A.inner_step
    A.inner_step = lambda idx, **kw: seen.append(list(idx)) or 0.0
    meta = Gecko(A, rng_mode="reference")
    random.seed(0)
    meta.train_step(tasks, num_shots=5, inner_batch_size=8, inner_iters=3, meta_step_size=0.0, meta_batch_size=1)
    random.seed(0)
    random.sample(list(tasks), 1)          # the task draw consumes the generator first (metaseg.py:244)
    exp = [list(b) for b in metaseg.mini_batch_indices(5, 8, 3)]
    assert seen == exp
    A.inner_step = orig


def test_gecko_two_steps_per_batch_quirk_and_fomlis_single():
    tasks = _tasks(1, 5)
    for cls, kw, expect in ((Gecko, {}, 2), (FOMLIS, dict(train_shots=5, tail_shots=None), 1)):
        A = _learner()
        calls = []
        A.inner_step = lambda idx, **k: calls.append(k.get("lr")) or 0.0
        m = cls(A, rng_mode="per_task", **kw)
        m.train_step(tasks, num_shots=5, inner_batch_size=4, inner_iters=2, meta_step_size=0.1, meta_batch_size=1, lr=0.5)
        assert len(calls) == 2 * expect          # Gecko runs two optimizer steps per batch when lr is given
        assert calls[0] == 0.5


def test_augmented_inner_loop_feeds_the_reference_schedule_and_restores_state():
    """--augment: every inner step trains on a host-augmented batch drawn in the reference's generator order (Reptile forwards no
    aug_rate -> keep probability 1/7; FOMAML forwards its own and leaves the tail raw); evaluation fine-tunes on augmented
    copies, predicts on the ORIGINAL test images and restores every variable."""
    from mliis_amd import augment
    tasks = _tasks(1, 6)
    x0, y0 = tasks[0].images.numpy(), tasks[0].labels.numpy()
    for cls, kw, fomaml in ((Gecko, {}, False), (FOMLIS, dict(train_shots=6, tail_shots=2), True)):
        A = _learner()
        fed = []
        orig_load = A.load_task
        A.load_task = lambda xs, ys: (fed.append((np.array(xs, dtype=np.float32), np.array(ys, dtype=np.float32))), orig_load(xs, ys))[1]
        augment._SHARED_ORDER[:] = augment.PRISTINE_ORDER
        random.seed(5)
        np.random.seed(6)
        m = cls(A, rng_mode="reference", augment=True, aug_rate=0.5, **kw)
        m.train_step(tasks, num_shots=6, inner_batch_size=4, inner_iters=3, meta_step_size=0.1, meta_batch_size=1)
        # replay: task draw, then the reference-order schedule with a fresh augmenter on the same streams
        augment._SHARED_ORDER[:] = augment.PRISTINE_ORDER
        random.seed(5)
        np.random.seed(6)
        random.sample(list(tasks), 1)
        exp = list(metaseg.augmented_batches(x0, y0, 4, 3, False, augment.Augmenter(verbose=False), 0.5 if fomaml else None,
                                             tail_shots=2 if fomaml else None, fomaml=fomaml))
        assert len(fed) == 3 == len(exp)
        for (fx, fy), (ex, ey) in zip(fed, exp):
            assert np.array_equal(fx, ex) and np.array_equal(fy, ey)
        if fomaml:   # the last batch is the raw tail
            assert fed[-1][0].shape[0] == 2 and any(np.array_equal(fed[-1][0][0], x0[i]) for i in range(6))
    # evaluation
    A = _learner()
    before = (A.export_trainable().clone(), A.export_bn().clone())
    m = Gecko(A, rng_mode="reference", augment=True, aug_rate=0.5)
    random.seed(1)
    np.random.seed(2)
    iou, per_task = m.evaluate(list(tasks), num_shots=4, inner_batch_size=4, inner_iters=2, test_shots=2)
    assert 0.0 <= iou <= 1.0 and list(per_task) == ["t0"]
    assert torch.equal(A.export_trainable(), before[0]) and torch.equal(A.export_bn(), before[1])


def test_fomlis_sample_train_val_with_replacement():
    """--sample_foml_train_val_with_replacement (reptile.py:657-658): in reference mode the head / tail come from the global numpy
    generator right after the task draw; in per-task mode from a private stream (rank-count independent, reproducible)."""
    tasks = _tasks(2, 6)
    A = _learner()
    seen = []
    A.inner_step = lambda idx, **kw: seen.append(list(idx)) or 0.0
    meta = FOMLIS(A, train_shots=6, tail_shots=2, sample_train_val_with_replacement=True, rng_mode="reference")
    assert meta.train_shots == 4
    random.seed(2)
    np.random.seed(3)
    meta.train_step(tasks, num_shots=6, inner_batch_size=3, inner_iters=3, meta_step_size=0.0, meta_batch_size=1)
    random.seed(2)
    np.random.seed(3)
    random.sample(list(tasks), 1)
    exp = metaseg.fomaml_batch_indices(6, 2, 3, 3, with_replacement_train_shots=4)
    assert seen == exp and len(seen[-1]) == 2
    runs = []
    for _ in range(2):
        seen.clear()
        m = FOMLIS(A, train_shots=6, tail_shots=2, sample_train_val_with_replacement=True, rng_mode="per_task", seed=8)
        m.train_step(tasks, num_shots=6, inner_batch_size=3, inner_iters=3, meta_step_size=0.0, meta_batch_size=2)
        runs.append([list(b) for b in seen])
    assert runs[0] == runs[1] and len(runs[0]) == 6
    with pytest.raises(ValueError):
        FOMLIS(A, train_shots=5, tail_shots=None, sample_train_val_with_replacement=True)


def test_unsupported_features_raise():
    A = _learner()
    with pytest.raises(ValueError):
        Gecko(A, rng_mode="reference", dist=type("D", (), {"rank": 0, "world": 2})())


# ------------------------------------------------------------------------------------------------ gloo, world size 2
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fomaml, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = _learner()
        tasks = _tasks(3, 6)
        meta = FOMLIS(A, train_shots=6, tail_shots=2, seed=1) if fomaml else Gecko(A, seed=1)
        assert meta.rng_mode == "per_task" and meta.dist.world == world
        for _ in range(2):
            meta.train_step(tasks, num_shots=6, inner_batch_size=4, inner_iters=2, replacement=False, meta_step_size=0.3, meta_batch_size=4)
        q.put((rank, A.export_trainable().numpy(), A.export_bn().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fomaml", [False, True])
def test_gloo_world2_equals_single_process(fomaml):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fomaml, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # every rank ends with the same parameters
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])
    # and they equal the single-process (world 1, per-task rng) run
    torch.set_num_threads(4)
    A = _learner()
    tasks = _tasks(3, 6)
    meta = FOMLIS(A, train_shots=6, tail_shots=2, seed=1, rng_mode="per_task") if fomaml else Gecko(A, seed=1, rng_mode="per_task")
    for _ in range(2):
        meta.train_step(tasks, num_shots=6, inner_batch_size=4, inner_iters=2, replacement=False, meta_step_size=0.3, meta_batch_size=4)
    np.testing.assert_allclose(res[0][1], A.export_trainable().numpy(), rtol=0, atol=1e-9)  # thread-count dependent fp64 rounding
    np.testing.assert_allclose(res[0][2], A.export_bn().numpy(), rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("fomaml", [False, True])
def test_concurrent_lanes_host_logic_equals_task_by_task(fomaml):
    """Gecko(lanes=...) -- tasks adapted several at a time on learners of their own (on the GPU: own streams) -- gives the update of
    the task-by-task loop exactly, for training (5 tasks on 1 + 2 learners: a full group and a ragged one) and for evaluate()."""
    tasks = _tasks(4, 6)
    res = []
    for n_lanes in (0, 2):
        A = _learner()
        lanes = [R.OracleLearner(image_size=H, seed=90 + k, dtype=torch.float64, lr=1e-2, drop_connect=False) for k in range(n_lanes)]
        kw = dict(rng_mode="per_task", seed=4, lanes=lanes)
        meta = FOMLIS(A, train_shots=6, tail_shots=2, **kw) if fomaml else Gecko(A, **kw)
        for _ in range(2):
            meta.train_step(tasks, num_shots=6, inner_batch_size=4, inner_iters=3, meta_step_size=0.5, meta_batch_size=5)
        random.seed(5)
        ev = meta.evaluate(list(tasks), num_shots=3, test_shots=3, inner_batch_size=2, inner_iters=2, eval_all_tasks=True)
        res.append((A.export_trainable().clone(), A.export_bn().clone(), ev, random.random()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] and res[0][3] == res[1][3]
