"""Generates tests/golden/host_logic.json by IMPORTING the reference's own pure-numpy/Python host logic
(from /root/reference, read-only) under stub `tensorflow` / `skopt` / `imageio` / `matplotlib` modules.

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_host_logic_golden.py

Only inputs and outputs (data) are stored; no reference source text is written anywhere.
"""
import importlib
import json
import os
import random
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_logic.json")


class _Anything(types.ModuleType):
    """A module whose every attribute is another _Anything / a permissive callable class."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if name[:1].isupper():
            cls = type(name, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, *a, **k: None})
            setattr(self, name, cls)
            return cls
        m = _Anything(full)
        m.__call__ = lambda *a, **k: None
        sys.modules[full] = m
        setattr(self, name, m)
        return m

    def __call__(self, *a, **k):
        return None


def _install_stubs():
    for root in ("tensorflow", "skopt", "imageio", "matplotlib", "pandas_stub_unused"):
        sys.modules[root] = _Anything(root)
    for sub in ("tensorflow.contrib", "tensorflow.contrib.tpu", "tensorflow.contrib.tpu.python",
                "tensorflow.contrib.tpu.python.ops", "tensorflow.contrib.tpu.python.ops.tpu_ops",
                "tensorflow.contrib.tpu.python.tpu", "tensorflow.contrib.tpu.python.tpu.tpu_function",
                "tensorflow.python", "tensorflow.python.tpu", "tensorflow.python.tpu.tpu_function",
                "tensorflow.python.tpu.ops", "tensorflow.python.tpu.ops.tpu_ops",
                "skopt.space", "matplotlib.pyplot", "matplotlib.patches"):
        sys.modules[sub] = _Anything(sub)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)


def main():
    _install_stubs()
    G = {}

    lrs = importlib.import_module("models.lr_schedulers")
    G["cosine"] = {"args": [0.005, 8], "lr": [float(lrs.CosineLRScheduler(0.005, 8).cur_lr(t)) for t in range(12)]}
    G["step"] = {"args": [0.005, 8, 0.5, 5], "lr": [float(lrs.StepDecay(0.005, 8, 0.5, 5).cur_lr(t)) for t in range(40)]}
    G["step_b"] = {"args": [0.01, None, 0.1, 2], "lr": [float(lrs.StepDecay(0.01, None, 0.1, 2).cur_lr(t)) for t in range(14)]}
    G["schedulers"] = sorted(k for k in lrs.supported_learning_rate_schedulers)

    V = importlib.import_module("meta_learners.variables")
    rng = np.random.default_rng(7)
    old = [rng.standard_normal(s).astype(np.float32) for s in [(3, 2), (5,), (2, 2, 2)]]
    news = [[rng.standard_normal(a.shape).astype(np.float32) for a in old] for _ in range(3)]
    G["vars"] = {
        "old": [a.tolist() for a in old],
        "news": [[a.tolist() for a in n] for n in news],
        "average": [a.tolist() for a in V.average_vars(news)],
        "interpolate_0.1": [a.tolist() for a in V.interpolate_vars(old, V.average_vars(news), 0.1)],
        "subtract": [a.tolist() for a in V.subtract_vars(news[0], old)],
        "add_scaled_0.25": [a.tolist() for a in V.add_vars(old, V.scale_vars(V.subtract_vars(news[1], old), 0.25))],
        "interp_simple": [float(x) for x in V.interpolate_vars([np.float32(0), np.float32(1), np.float32(2), np.float32(3)],
                                                             [np.float32(0), np.float32(3), np.float32(6), np.float32(9)], 0.1)],
    }

    MS = importlib.import_module("meta_learners.metaseg")
    cases = []
    for (n, bs, nb, seed) in [(5, 8, 3, 0), (5, 8, 8, 1), (10, 8, 7, 2), (3, 2, 5, 3), (8, 8, 2, 4), (1, 4, 2, 5)]:
        random.seed(seed)
        out = [[s[0] for s in b] for b in MS._mini_batches([(i, i) for i in range(n)], bs, nb, False)]
        cases.append({"n": n, "batch": bs, "num_batches": nb, "seed": seed, "replacement": False, "batches": out})
    for (n, bs, nb, seed) in [(10, 4, 3, 0), (6, 6, 2, 9)]:
        random.seed(seed)
        out = [[s[0] for s in b] for b in MS._mini_batches([(i, i) for i in range(n)], bs, nb, True)]
        cases.append({"n": n, "batch": bs, "num_batches": nb, "seed": seed, "replacement": True, "batches": out})
    G["mini_batches"] = cases
    splits = []
    for (n, ts, seed) in [(10, 5, 0), (10, 1, 1), (6, 5, 2), (15, 5, 3)]:
        random.seed(seed)
        tr, te = MS._split_train_test_segmentation([(i, i) for i in range(n)], ts)
        splits.append({"n": n, "test_shots": ts, "seed": seed, "train": [s[0] for s in tr], "test": [s[0] for s in te]})
    G["split"] = splits
    # FOMAML batch schedule (FOMLIS._mini_batches, reptile.py:649-663) = split + (inner_iters-1) batches + tail
    foml = []
    for (n, tail, bs, iters, seed) in [(10, 5, 8, 8, 0), (10, 5, 8, 3, 11), (7, 2, 4, 5, 5)]:
        random.seed(seed)
        tr, te = MS._split_train_test_segmentation([(i, i) for i in range(n)], tail)
        bt = [[s[0] for s in b] for b in MS._mini_batches(tr, bs, iters - 1, False)]
        bt.append([s[0] for s in te])
        foml.append({"n": n, "tail": tail, "batch": bs, "inner_iters": iters, "seed": seed, "batches": bt})
    G["foml_batches"] = foml
    # --sample_foml_train_val_with_replacement (reptile.py:657-658; metaseg.py:313-318): numpy draws, then the head schedule, then the tail
    wr = []
    for (n, train, tail, bs, iters, seed) in [(10, 5, 5, 8, 8, 0), (10, 5, 5, 8, 3, 4), (6, 4, 2, 4, 4, 7)]:
        random.seed(seed)
        np.random.seed(seed + 100)
        tr, te = MS._sample_train_test_segmentation_with_replacement([(i, i) for i in range(n)], train_shots=train, test_shots=tail)
        bt = [[s[0] for s in b] for b in MS._mini_batches(tr, bs, iters - 1, False)]
        bt.append([s[0] for s in te])
        wr.append({"n": n, "train": train, "tail": tail, "batch": bs, "inner_iters": iters, "seed": seed, "np_seed": seed + 100,
                   "head": [s[0] for s in tr], "batches": bt})
    G["foml_with_replacement"] = wr

    R = importlib.import_module("meta_learners.supervised_reptile.supervised_reptile.reptile")
    rng = np.random.default_rng(3)
    ious = []
    for _ in range(6):
        h, w = int(rng.integers(3, 9)), int(rng.integers(3, 9))
        p1 = (rng.random((h, w)) > 0.5).astype(np.float32)
        l1 = rng.random((h, w)).astype(np.float32)      # fractional labels get rounded
        pred = np.stack([1 - p1, p1], -1)
        lab = np.stack([1 - l1, l1], -1)
        ious.append({"pred": pred.tolist(), "label": lab.tolist(), "iou": float(R.Gecko._iou(pred, lab))})
    z = np.zeros((4, 4, 2), np.float32)
    ious.append({"pred": z.tolist(), "label": z.tolist(), "iou": float(R.Gecko._iou(z, z))})
    G["iou"] = ious
    meas = []
    for _ in range(3):
        y = rng.random((6, 6)).astype(np.float32)
        p = rng.random((6, 6)).astype(np.float32)
        tp, tn, fp, fn = R.measure(y, p)
        meas.append({"y": y.tolist(), "pred": p.tolist(), "tp": int(tp), "tn": int(tn), "fp": int(fp), "fn": int(fn),
                     "iou_img": float(R.iou_img(tp, fp, fn))})
    G["measure"] = meas

    HS = importlib.import_module("meta_learners.hyperparam_search")
    es_cases = []
    for (pat, mn, seq) in [(2, 1, [0.1, 0.3, 0.2, 0.25, 0.29, 0.1]), (0, 0, [0.5, 0.4, 0.6]), (3, 4, [0.1, 0.2, 0.3, 0.2, 0.1, 0.1, 0.1, 0.1, 0.1])]:
        es = HS.EarlyStopper(patience=pat, min_steps=mn)
        cont = [bool(es.continue_training(m, i + 1)) for i, m in enumerate(seq)]
        es_cases.append({"patience": pat, "min_steps": mn, "metrics": seq, "continue": cont,
                         "best_num_steps": es.best_num_steps(), "best_metric": es.best_metric()})
    G["early_stopper"] = es_cases

    F = importlib.import_module("data.fss_1000_utils")
    G["fss_test_tasks"] = list(F.TEST_TASK_IDS)
    G["fss_n_train"] = len(getattr(F, "TRAIN_TASK_IDS", [])) or None
    G["fp_k_test_tasks"] = list(getattr(F, "FP_K_TEST_TASK_IDS", []))

    U = importlib.import_module("utils.util")
    G["ci95"] = [{"a": a, "v": float(U.ci95(a))} for a in ([0.1, 0.5, 0.9, 0.3], [0.7] * 5, [0.2, 0.8])]

    A = importlib.import_module("meta_learners.args")
    p = A.argument_parser()
    d0 = vars(p.parse_args([]))
    d1 = vars(p.parse_args("--rsd 2 4 --sgd --foml --foml-tail 5 --image_size 224 --l2 --loss_name bce_dice".split()))
    G["argparse_defaults"] = {k: v for k, v in d0.items() if isinstance(v, (int, float, str, bool, type(None), list))}
    G["argparse_runsh_like"] = {k: v for k, v in d1.items() if isinstance(v, (int, float, str, bool, type(None), list))}
    # meta-step-size anneal (train.py:90-92) is inline code, restated: values for reference
    with open(OUT, "w") as f:
        json.dump(G, f, indent=1, sort_keys=True)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
