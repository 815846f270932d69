"""Generates tests/golden/hyperparam_search.json by EXECUTING the reference's own definitions (run here, where /root/reference is
mounted; the fixture holds inputs and outputs only).  The module itself cannot be imported -- it needs scikit-optimize at import time --
so the four pure-Python definitions under test are compiled from its syntax tree.

    python tests/golden/make_hyperparam_golden.py
"""
import ast
import contextlib
import io
import json
import operator
import os
import tempfile
import typing

import numpy as np
import pandas as pd

REF = "/root/reference/meta_learners/hyperparam_search.py"
WANT = {"EarlyStopper", "run_m", "save_results", "compute_best_configuration"}


def load():
    tree = ast.parse(open(REF).read())
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in WANT]
    ns = {"operator": operator, "np": np, "pd": pd, "os": os}
    ns.update({k: getattr(typing, k) for k in ("Callable", "Dict", "List", "Tuple", "Optional", "Any")})
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns


def main():
    ns = load()
    out = {"stopper": [], "best": [], "csv": {}}
    rs = np.random.RandomState(0)
    with contextlib.redirect_stdout(io.StringIO()):
        for patience, inc, min_steps, n in [(3, True, 0, 30), (0, True, 0, 10), (2, False, 0, 25), (5, True, 4, 40), (1, True, 10, 12),
                                            (50, True, 1, 80), (2, True, 3, 20)]:
            metrics = [float(x) for x in np.round(rs.rand(n) + np.linspace(0, 0.5 if inc else -0.5, n), 3)]
            if n == 10:
                metrics[3] = float("nan")
            st = ns["EarlyStopper"](patience, metric_should_increase=inc, min_steps=min_steps)
            flags = []
            for i, m in enumerate(metrics):
                flags.append(bool(st.continue_training(m, i + 1)))
                if not flags[-1]:
                    break
            out["stopper"].append({"patience": patience, "increase": inc, "min_steps": min_steps, "metrics": metrics, "flags": flags,
                                   "best_num_steps": st.best_num_steps(), "best_metric": st.best_metric()})
        results = [({"lr": 0.01, "inner_batch_size": 8}, (["a", "b", "c"], [3, 9, 4], [0.5, 0.7, 0.6])),
                   ({"lr": 0.002, "inner_batch_size": 4}, (["a", "b", "c"], [10, 12, 30], [0.65, 0.7, 0.62])),
                   ({"lr": 0.03, "inner_batch_size": 5}, (["a", "b"], [1, 2], [0.2, 0.9]))]
        for inc in (True, False):
            cfg, steps, metric = ns["compute_best_configuration"](results, metric_should_increase=inc)
            out["best"].append({"increase": inc, "config": cfg, "steps": steps, "metric": float(metric)})
        calls = iter(results)
        ids, steps, mets = ns["run_m"](lambda **kw: next(calls)[1], {}, m=2)
        out["run_m"] = {"ids": ids, "steps": steps, "metrics": mets}
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "r.csv")
            ns["save_results"](results[:1], p)
            ns["save_results"](results[1:2], p, append_if_exists=True)
            ns["save_results"](results[2:], p)                      # exists, no append -> r.csv_0
            ns["save_results"](results[:2], p)                      # -> r.csv_1
            for name in sorted(os.listdir(d)):
                out["csv"][name] = open(os.path.join(d, name)).read()
    out["results"] = [[c, list(r)] for c, r in results]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hyperparam_search.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
