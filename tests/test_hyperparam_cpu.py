"""Update-hyperparameter search harness: the early-stopping / result-table logic against goldens produced by
executing the reference's own definitions, the GP / expected-improvement optimiser on known objectives, and the early-stopping and
k-shot evaluation loops of Gecko on the CPU oracle learner."""
import contextlib
import io
import json
import math
import os
import random

import numpy as np
import pytest
import torch

from mliis_amd import hyperparam_search as hs
from mliis_amd import metaseg
from mliis_amd.reptile import Gecko

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hyperparam_search.json")))


def _same(a, b):
    return (a is None and b is None) or (isinstance(a, float) and isinstance(b, float) and math.isnan(a) and math.isnan(b)) or a == b


def test_early_stopper_matches_reference_traces():
    for case in GOLD["stopper"]:
        st = hs.EarlyStopper(case["patience"], metric_should_increase=case["increase"], min_steps=case["min_steps"])
        flags = []
        for i, m in enumerate(case["metrics"]):
            flags.append(st.continue_training(m, i + 1))
            if not flags[-1]:
                break
        assert flags == case["flags"]
        assert _same(st.best_num_steps(), case["best_num_steps"]) and _same(st.best_metric(), case["best_metric"])


def test_result_table_and_best_configuration_match_reference(tmp_path):
    results = [(c, tuple(r)) for c, r in GOLD["results"]]
    for case in GOLD["best"]:
        cfg, steps, metric = hs.compute_best_configuration(results, metric_should_increase=case["increase"])
        assert cfg == case["config"] and steps == case["steps"] and metric == pytest.approx(case["metric"], abs=0, rel=1e-15)
    calls = iter(results)
    ids, steps, mets = hs.run_m(lambda **kw: next(calls)[1], {}, m=2)
    assert (ids, steps, mets) == (GOLD["run_m"]["ids"], GOLD["run_m"]["steps"], GOLD["run_m"]["metrics"])
    p = str(tmp_path / "r.csv")
    hs.save_results(results[:1], p)
    hs.save_results(results[1:2], p, append_if_exists=True)
    hs.save_results(results[2:], p)
    hs.save_results(results[:2], p)
    assert {n: open(os.path.join(str(tmp_path), n)).read().replace("\r\n", "\n") for n in sorted(os.listdir(str(tmp_path)))} == GOLD["csv"]


def test_dimension_transforms_and_types():
    d = hs.Dimension(0.0005, 0.05, "lr")
    assert d.from_unit(0.0) == pytest.approx(0.0005) and d.from_unit(1.0) == pytest.approx(0.05)
    assert d.from_unit(0.5) == pytest.approx(math.sqrt(0.0005 * 0.05))          # log-uniform: the midpoint is the geometric mean
    assert d.to_unit(d.from_unit(0.3)) == pytest.approx(0.3)
    i = hs.Dimension(2, 32, "inner_batch_size", integer=True)
    vals = {i.from_unit(u) for u in np.linspace(0, 1, 101)}
    assert all(isinstance(v, int) and 2 <= v <= 32 for v in vals) and {2, 32} <= vals
    assert hs.get_dim_type([0.1, 0.2]) == "real" and hs.get_dim_type([4, 8]) == "integer"
    with pytest.raises(ValueError):
        hs.Dimension(0.5, 0.5, "x")
    with pytest.raises(ValueError):
        hs.get_dim_type([None, None])


def test_gp_optimizer_finds_the_optimum_of_a_known_objective():
    np.random.seed(0)
    dims = [hs.Dimension(0.0005, 0.05, "lr"), hs.Dimension(0.1, 0.8, "drop_rate")]
    f = lambda lr, dr: (math.log2(lr) - math.log2(0.004)) ** 2 + 4.0 * (math.log2(dr) - math.log2(0.3)) ** 2   # noqa: E731
    opt = hs.GPOptimizer(dims, n_initial_points=8, n_candidates=2000)
    vals = []
    for _ in range(24):
        x = opt.ask()
        assert 0.0005 <= x[0] <= 0.05 and 0.1 <= x[1] <= 0.8
        vals.append(f(*x))
        opt.tell(x, vals[-1])
    assert min(vals[8:]) < min(vals[:8])            # the model-guided phase improves on the random phase
    assert min(vals) < 0.05                         # and lands next to the optimum (objective range ~ 0 .. 60)
    best = opt.X[int(np.argmin(vals))]
    assert abs(dims[0].from_unit(best[0]) / 0.004 - 1) < 0.25


def test_gp_search_driver_writes_the_table_and_returns_the_best_lr(tmp_path):
    np.random.seed(1)
    seen = []

    def eval_fn(dataset, lr, drop_rate, aug_rate, inner_batch_size, **kw):
        seen.append((lr, drop_rate, aug_rate, inner_batch_size))
        assert isinstance(inner_batch_size, int) and 4 <= inner_batch_size <= 16 and drop_rate == 0.2 and aug_rate == 0.5
        score = 1.0 - abs(math.log10(lr) - math.log10(0.01))            # best mIoU at lr = 0.01
        return ["t0", "t1"], [5, 7], [score, score - 0.1]

    # single-point ranges (drop rate, aug rate here) are not searched: those keys keep the values preset in params
    params = {"dataset": None, "lr": None, "drop_rate": 0.2, "aug_rate": 0.5, "inner_batch_size": 8}
    out = str(tmp_path / "uho.csv")
    with contextlib.redirect_stdout(io.StringIO()):
        lr, steps = hs.lr_droprate_aug_rate_batch_size_gp_search(eval_fn, params, lr_search_range_low=0.05, lr_search_range_high=0.0005,
                                                                 batch_size_search_range_low=4, batch_size_search_range_high=16,
                                                                 n=12, m=2, save_results_to=out)
    assert len(seen) == 24 and steps == 6                                # n * m evaluations; median of [5, 7, 5, 7]
    assert 0.004 < lr < 0.025
    rows = open(out).read().strip().split("\n")
    assert rows[0] == "task_ID,best_num_steps,mIoU,lr,inner_batch_size" and len(rows) == 1 + 12 * 4
    # a fixed lr range is not searched and comes back unchanged
    with contextlib.redirect_stdout(io.StringIO()):
        lr, _ = hs.lr_droprate_aug_rate_batch_size_gp_search(lambda **kw: (["t"], [3], [0.5]), dict(params), lr_search_range_low=0.01,
                                                             lr_search_range_high=0.01, n=2, save_results_to=None)
    assert lr == 0.01


# ---------------------------------------------------------------------------------------------------- Gecko harness on the oracle
H = 32


def _learner():
    from oracle import efficientlab_ref as R
    return R.OracleLearner(image_size=H, seed=3, dtype=torch.float64, lr=5e-3, drop_connect=False)


def _task(n, seed, name="t"):
    x, y = metaseg.synthetic_task(n, H, seed=seed, block=4)
    return metaseg.DeviceTask(name, torch.tensor(x), torch.tensor(y))


def test_early_stopping_learn_follows_the_stopper_and_restores_state():
    from mliis_amd.metrics import iou
    L = _learner()
    task = _task(7, 11)
    g = Gecko(L, transductive=True, rng_mode="reference")
    L.load_task(task.images, task.labels)
    tr, va = [0, 1, 2, 3], [4, 5, 6]
    before = (L.export_trainable().clone(), L.export_bn().clone())
    random.seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        steps, best = g._early_stopping_learn(tr, va, task.labels, 4, min_steps=1, max_steps=9, replacement=False, lr=5e-3, patience=2)
    assert torch.equal(L.export_trainable(), before[0]) and torch.equal(L.export_bn(), before[1])
    # independent replay: same batches, a metric after every step, the reference's stopping rule
    random.seed(3)
    st = hs.EarlyStopper(2, min_steps=1)
    lab = task.labels.numpy()
    for it, b in enumerate(metaseg.mini_batch_indices(4, 4, 9, False)):
        L.inner_step([tr[i] for i in b], lr=5e-3)
        preds = L.predict_resident(va, training=False).cpu().numpy()
        m = np.nanmean([iou(preds[j], lab[va[j]]) for j in range(3)])
        if not st.continue_training(m, it + 1):
            break
    assert (steps, best) == (st.best_num_steps(), st.best_metric()) and 1 <= steps <= 9
    with pytest.raises(ValueError):
        g._early_stopping_learn(tr, va, task.labels, 4, 0, 2, False, lr_scheduler=object(), lr=0.1)


def test_evaluate_with_early_stopping_both_modes():
    L = _learner()
    tasks = [_task(6, 20, "a"), _task(6, 21, "b")]
    g = Gecko(L, transductive=True, rng_mode="reference")
    g.ES_PATIENCE = 1
    random.seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        names, steps, ious = g.evaluate_with_early_stopping(list(tasks), num_shots=4, inner_batch_size=4, min_steps=0, max_steps=4,
                                                            replacement=False, eval_all_tasks=True, test_shots=2, lr=5e-3)
    assert names == ["a", "b"] and len(steps) == 2 == len(ious) and all(1 <= s <= 4 for s in steps) and all(0 <= v <= 1 for v in ious)
    # min_steps == max_steps: no early stopping, plain evaluation with that many steps
    calls = []
    orig = L.inner_step
    L.inner_step = lambda idx, **kw: calls.append(kw) or orig(idx, **kw)
    with contextlib.redirect_stdout(io.StringIO()):
        names, steps, ious = g.evaluate_with_early_stopping(list(tasks), num_shots=4, inner_batch_size=4, min_steps=3, max_steps=3,
                                                            replacement=False, eval_all_tasks=True, test_shots=2, lr=5e-3, drop_rate=0.25)
    assert steps == [3, 3] and len(calls) == 6 and all(k["lr"] == 5e-3 and k["drop_rate"] == 0.25 for k in calls)
    assert sorted(names) == ["a", "b"] and len(ious) == 2


def test_k_shot_range_uses_early_stopping_from_ten_shots_on():
    L = _learner()
    task = _task(14, 30, "k")
    g = Gecko(L, transductive=True, rng_mode="reference")
    g.ES_PATIENCE, g.K_SHOT_ES_MAX_STEPS = 0, 3
    before = L.export_trainable().clone()
    es_calls = []
    orig = g._early_stopping_learn
    g._early_stopping_learn = lambda tr, va, *a, **k: es_calls.append((len(tr), len(va), k["min_steps"], k["max_steps"])) or orig(tr, va, *a, **k)
    random.seed(2)
    with contextlib.redirect_stdout(io.StringIO()):
        ks, res = g.evaluate_m_k_shot_ranges_all_tasks([task], k_range=[1, 3, 10], m=2, inner_batch_size=4, inner_iters=2, replacement=False,
                                                       lr=5e-3, test_samples=4, iter_range=[1, 1, 1], aug_rate=0.5)
    assert ks == [1, 3, 10, 1, 3, 10] and len(res) == 6 and all(0 <= v <= 1 for v in res)
    assert es_calls == [(8, 2, 1, 3), (8, 2, 1, 3)]                    # k = 10: 20 % held out for early stopping; k < 10: none
    assert torch.equal(L.export_trainable(), before)
