"""Update-hyperparameter optimisation (UHO) around the inner loop: early stopping, the per-configuration evaluation driver, the
result table and the Bayesian search over (lr, final-layer drop rate, augmentation rate, inner batch size).

Restates meta_learners/hyperparam_search.py of the reference:
  EarlyStopper                                :24-69     (bit-for-bit: pinned by tests/golden/hyperparam_search.json, produced by
                                                          executing the reference's own class)
  run_m / save_results / compute_best_configuration / log_opt_progress    :72-163 (same csv columns, same file-collision rule)
  gp_update_hyperparameter_optimization       :182-249
  lr_droprate_aug_rate_batch_size_gp_search   :252-281

The reference delegates the search itself to scikit-optimize (`skopt.Optimizer(dims, "GP", acq_func="EI", acq_optimizer="lbfgs",
n_initial_points=n/2)`, requirements.txt:7, version not pinned), which is not vendored in the reference tree and not installed
here.  `GPOptimizer` below restates that published algorithm with numpy/scipy -- log-uniform (base 2) search dimensions normalised
to the unit cube, random initial points, then a Matern-5/2 Gaussian process (ARD length scales + noise, fitted by maximising the
marginal likelihood) whose expected improvement is maximised from the best of 10000 random candidates with L-BFGS-B.  It follows
the same ask/tell protocol and consumes `numpy.random` (seed it for reproducible searches), but it cannot draw the SAME
configurations as skopt: this part of the harness is behaviourally, not numerically, pinned (tests: it finds the optimum of known
objectives and respects the bounds / types).
"""
from __future__ import annotations

import csv
import operator
import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

DROPOUT_RATE_NAME = "drop_rate"
AUG_RATE_NAME = "aug_rate"
BATCH_SIZE_NAME = "inner_batch_size"
LEARNING_RATE_NAME = "lr"
SUPPORTED_SEARCH_ALGS = {"GP"}


class EarlyStopper:
    """Stopping criterion from a metric and a patience: training continues while the metric improved within the last `patience`
    evaluations; evaluations at or before `min_steps` only record the metric."""

    def __init__(self, patience: int = 10, metric_should_increase: bool = True, min_steps: int = 0):
        self.patience = patience
        self.metric_should_increase = metric_should_increase
        self.eval_operator = operator.gt if metric_should_increase else operator.lt
        self._best_metric = None
        self._best_num_steps = min_steps if min_steps > 0 else None
        self.num_evals_without_improving = 0
        self.min_steps = min_steps
        print("Built EarlyStopper with patience {}".format(self.patience))

    def continue_training(self, metric, total_steps_taken) -> bool:
        if total_steps_taken <= self.min_steps:
            self._best_metric = metric
            return True
        if self._best_metric is None or self.eval_operator(metric, self._best_metric):
            self.num_evals_without_improving = 0
            self._best_metric = metric
            self._best_num_steps = total_steps_taken
            return True
        self.num_evals_without_improving += 1
        return self.num_evals_without_improving <= self.patience

    def best_metric(self):
        return self._best_metric

    def best_num_steps(self):
        return self._best_num_steps


def run_m(eval_fn: Callable, params: Dict, m: int = 1):
    """m calls of eval_fn(**params) -> concatenated (task ids, best step counts, metrics)."""
    ids, steps, metrics = [], [], []
    for _ in range(m):
        a, b, c = eval_fn(**params)
        ids.extend(a)
        steps.extend(b)
        metrics.extend(c)
    return ids, steps, metrics


def save_results(results: List[Tuple[Dict, Tuple[List, List, List]]], path: str, metric_name: str = "mIoU", append_if_exists: bool = False):
    """One csv row per (configuration, task): columns task_ID, best_num_steps, <metric>, then the configuration keys in first-seen
    order.  An existing file is appended to (no header) or, without append_if_exists, left alone and `path_<i>` is written."""
    print("Saving results to {}".format(path))
    table: Dict[str, list] = {"task_ID": [], "best_num_steps": [], metric_name: []}
    for config, (task_ids, num_steps, metrics) in results:
        for key, val in config.items():
            table.setdefault(key, []).extend([val] * len(task_ids))
        table["task_ID"].extend(task_ids)
        table["best_num_steps"].extend(num_steps)
        table[metric_name].extend(metrics)
    mode, header = "w", True
    if os.path.exists(path):
        if append_if_exists:
            mode, header = "a", False
        else:
            i = 0
            while os.path.exists(path + "_{}".format(i)):
                i += 1
            path = path + "_{}".format(i)
    cols = list(table)
    with open(path, mode, newline="") as f:
        w = csv.writer(f)
        if header:
            w.writerow(cols)
        for row in zip(*[table[c] for c in cols]):   # (missing values as empty cells, like the reference's DataFrame.to_csv)
            w.writerow(["" if isinstance(v, float) and v != v else v for v in row])
    print("Saved optimization raw results to {}".format(path))
    return path


def compute_best_configuration(results_list, metric_should_increase: bool = True):
    """Configuration with the best mean metric over its tasks, the median of its best step counts, and that mean."""
    better = operator.gt if metric_should_increase else operator.lt
    best_metric = -np.inf if metric_should_increase else np.inf
    best_config, best_step_num = None, None
    for sampled_config, (task_ids, num_steps, metrics) in results_list:
        mean_metric = np.mean(metrics)
        if better(mean_metric, best_metric):
            best_config, best_metric, best_step_num = sampled_config, mean_metric, np.median(num_steps)
    print("Best mIoU found: {}".format(best_metric))
    print("with median iteration: {}".format(best_step_num))
    print("and config: {}".format(best_config))
    return best_config, int(best_step_num), best_metric


def log_opt_progress(hyperparams, results_i, task_ids, num_steps, metrics, save_results_to):
    print("Results for hyperparams {}: task IDs: {}, best num steps: {}, mIoUs: {}".format(hyperparams, task_ids, num_steps, metrics))
    print("mean mIoU: {}".format(np.nanmean(metrics)))
    if save_results_to is not None:
        save_results([results_i], save_results_to, append_if_exists=True)


# ---------------------------------------------------------------------------------------------------- search space + GP / EI
class Dimension:
    """One search dimension on [low, high] (low < high), log-uniform in base `base` like skopt's Real/Integer(prior="log-uniform")."""

    def __init__(self, low, high, name: str, integer: bool = False, prior: str = "log-uniform", base: int = 2):
        if not low < high:
            raise ValueError("dimension {}: need low < high, got [{}, {}]".format(name, low, high))
        if prior not in ("log-uniform", "uniform"):
            raise ValueError("unknown prior {}".format(prior))
        if prior == "log-uniform" and low <= 0:
            raise ValueError("dimension {}: a log-uniform prior needs positive bounds".format(name))
        self.low, self.high, self.name, self.integer, self.log, self.base = low, high, name, integer, prior == "log-uniform", base

    def _fwd(self, v):
        return np.log(v) / np.log(self.base) if self.log else float(v)

    def to_unit(self, v) -> float:
        a, b = self._fwd(self.low), self._fwd(self.high)
        return (self._fwd(v) - a) / (b - a)

    def from_unit(self, u: float):
        a, b = self._fwd(self.low), self._fwd(self.high)
        t = a + min(1.0, max(0.0, float(u))) * (b - a)
        v = float(self.base) ** t if self.log else t
        if self.integer:
            return int(min(self.high, max(self.low, round(v))))
        return float(min(self.high, max(self.low, v)))


def get_dim_type(value: Sequence[Any]) -> str:
    v = value[0]
    if isinstance(v, bool) or not isinstance(v, (float, int, str)):
        raise ValueError("Value must be float, int, or str, but {} is {}".format(v, type(v)))
    if isinstance(v, str):
        raise ValueError("categorical dimensions are not used by the update-hyperparameter search")
    return "real" if isinstance(v, float) else "integer"


def _matern52(a, b, ls):
    d = np.sqrt(np.maximum(((a[:, None, :] - b[None, :, :]) / ls) ** 2, 0.0).sum(-1))
    s = np.sqrt(5.0) * d
    return (1.0 + s + s * s / 3.0) * np.exp(-s)


class GPOptimizer:
    """ask/tell minimiser: `n_initial_points` random configurations, then expected improvement under a Matern-5/2 GP."""

    def __init__(self, dims: Sequence[Dimension], n_initial_points: int = 10, xi: float = 0.01, n_candidates: int = 10000,
                 n_restarts: int = 5, rng=None):
        self.dims = list(dims)
        self.n_initial_points = max(1, int(n_initial_points))
        self.xi, self.n_candidates, self.n_restarts = xi, n_candidates, n_restarts
        self.rng = rng if rng is not None else np.random
        self.X: List[np.ndarray] = []   # unit-cube points told so far
        self.y: List[float] = []
        self._theta = None              # log [amplitude, noise, length scales...] of the last fit (warm start)

    # -- Gaussian process on the unit cube, targets standardised
    def _nll(self, theta, X, y):
        amp, noise, ls = np.exp(theta[0]), np.exp(theta[1]), np.exp(theta[2:])
        K = amp * _matern52(X, X, ls) + (noise + 1e-8) * np.eye(len(X))
        try:
            L = np.linalg.cholesky(K)
        except np.linalg.LinAlgError:
            return 1e25
        alpha = np.linalg.solve(L.T, np.linalg.solve(L, y))
        return 0.5 * y @ alpha + np.log(np.diag(L)).sum() + 0.5 * len(X) * np.log(2 * np.pi)

    def _fit(self):
        from scipy.optimize import minimize
        X = np.stack(self.X)
        y = np.asarray(self.y, dtype=np.float64)
        mu, sd = y.mean(), y.std()
        sd = sd if sd > 1e-12 else 1.0
        yn = (y - mu) / sd
        d = X.shape[1]
        bounds = [(-4.0, 4.0), (-12.0, 0.0)] + [(-4.0, 3.0)] * d
        starts = [np.concatenate([[0.0, -4.0], np.zeros(d) - 1.0])]
        if self._theta is not None:
            starts.append(self._theta)
        best = None
        for s in starts:
            r = minimize(self._nll, s, args=(X, yn), method="L-BFGS-B", bounds=bounds)
            if best is None or r.fun < best.fun:
                best = r
        self._theta = best.x
        amp, noise, ls = np.exp(best.x[0]), np.exp(best.x[1]), np.exp(best.x[2:])
        K = amp * _matern52(X, X, ls) + (noise + 1e-8) * np.eye(len(X))
        L = np.linalg.cholesky(K)
        alpha = np.linalg.solve(L.T, np.linalg.solve(L, yn))
        return X, L, alpha, amp, ls, yn.min()

    def _neg_ei(self, u, model):
        from scipy.stats import norm
        X, L, alpha, amp, ls, ybest = model
        u = np.atleast_2d(u)
        k = amp * _matern52(u, X, ls)
        mean = k @ alpha
        v = np.linalg.solve(L, k.T)
        std = np.sqrt(np.maximum(amp - (v * v).sum(0), 1e-12))
        imp = ybest - self.xi - mean
        z = imp / std
        return -(imp * norm.cdf(z) + std * norm.pdf(z))

    def ask(self) -> list:
        d = len(self.dims)
        if len(self.X) < self.n_initial_points:
            u = self.rng.uniform(0.0, 1.0, d)
        else:
            from scipy.optimize import minimize
            model = self._fit()
            cand = self.rng.uniform(0.0, 1.0, (self.n_candidates, d))
            vals = self._neg_ei(cand, model)
            order = np.argsort(vals)[: self.n_restarts]
            u, best = cand[order[0]], vals[order[0]]
            for i in order:
                r = minimize(lambda p: float(self._neg_ei(p, model)[0]), cand[i], method="L-BFGS-B", bounds=[(0.0, 1.0)] * d)
                if r.fun < best:
                    u, best = np.clip(r.x, 0.0, 1.0), r.fun
        return [dim.from_unit(x) for dim, x in zip(self.dims, u)]

    def tell(self, point: Sequence, objective: float):
        self.X.append(np.array([dim.to_unit(v) for dim, v in zip(self.dims, point)], dtype=np.float64))
        self.y.append(float(objective))
        return self


def insert_sampled_into_full_set_of_hyperparams(sampled: Dict, hyperparams: Dict) -> Dict:
    hyperparams.update(sampled)
    return hyperparams


def gp_update_hyperparameter_optimization(eval_fn: Callable, hyperparams: Dict, search_key_ranges: Dict, n: int,
                                          save_results_to: Optional[str] = "gp_hyper_param_search_results.csv", m: int = 1,
                                          metric_should_increase: bool = True, metric_name: str = "mIoU", base: int = 2,
                                          n_initial_points: Optional[int] = None, prior: str = "log-uniform"):
    """Multitask hyperparameter search: n configurations; each is evaluated m times with eval_fn(**hyperparams) (the sampled
    values replace the keys of `search_key_ranges` whose range is not a single point) and the negated mean metric is told to the
    optimiser.  Returns (best configuration, median best step count, best mean metric, all results)."""
    for key in search_key_ranges:
        assert key in hyperparams, "key: {} not in hyperparams: {}".format(key, hyperparams)
    if n_initial_points is None:
        n_initial_points = int(n / 2)
    print("Sampling {} points initially at random.".format(n_initial_points))
    dims = [Dimension(dom[0], dom[1], key, integer=get_dim_type(dom) == "integer", prior=prior, base=base)
            for key, dom in search_key_ranges.items() if dom[0] != dom[1]]
    opt = GPOptimizer(dims, n_initial_points=n_initial_points) if dims else None
    results = []
    for i in range(n):
        print("Running configuration sample {} of {}.".format(i + 1, n))
        print("With sampled hyperparams:")
        sampled_list = opt.ask() if opt is not None else []
        sampled = {dim.name: x for dim, x in zip(dims, sampled_list)}
        print(sampled)
        hyperparams = insert_sampled_into_full_set_of_hyperparams(sampled, hyperparams)
        task_ids, num_steps, metrics = run_m(eval_fn, hyperparams, m)
        objective = np.nanmean(metrics)
        if metric_should_increase:
            objective *= -1
        print("Objective value at sample {} of {}: {}".format(i + 1, n, objective))
        if opt is not None:
            opt.tell(sampled_list, objective)
        results_i = (sampled, (task_ids, num_steps, metrics))
        results.append(results_i)
        log_opt_progress(hyperparams, results_i, task_ids, num_steps, metrics, save_results_to)
    best_config, expected_best_step_num, best_metric = compute_best_configuration(results, metric_should_increase)
    return best_config, expected_best_step_num, best_metric, results


def lr_droprate_aug_rate_batch_size_gp_search(eval_fn: Callable, params: Dict, lr_name: str = LEARNING_RATE_NAME,
                                              lr_search_range_low: float = 0.0005, lr_search_range_high: float = 0.05,
                                              droprate_name: str = DROPOUT_RATE_NAME, drop_rate_search_range_low: float = 0.2,
                                              drop_rate_search_range_high: float = 0.2, aug_rate_name: str = AUG_RATE_NAME,
                                              aug_rate_search_range_low: float = 0.5, aug_rate_search_range_high: float = 0.5,
                                              batch_size_name: str = BATCH_SIZE_NAME, batch_size_search_range_low: int = 8,
                                              batch_size_search_range_high: int = 8, n: int = 100,
                                              save_results_to: str = "hyper_param_search_results.csv", m: int = 1,
                                              metric_should_increase: bool = True, metric_name: str = "mIoU") -> Tuple[float, int]:
    """Search over (lr, drop rate, aug rate, batch size); ranges given high-to-low are swapped, single-point ranges stay fixed.
    Returns (best lr -- the value in `params` when lr was not searched --, expected number of inner iterations)."""
    def rng(lo, hi, cast):
        lo, hi = cast(lo), cast(hi)
        return [hi, lo] if lo > hi else [lo, hi]

    ranges = {lr_name: rng(lr_search_range_low, lr_search_range_high, float),
              droprate_name: rng(drop_rate_search_range_low, drop_rate_search_range_high, float),
              aug_rate_name: rng(aug_rate_search_range_low, aug_rate_search_range_high, float),
              batch_size_name: rng(batch_size_search_range_low, batch_size_search_range_high, int)}
    best_config, expected_best_step_num, _, _ = gp_update_hyperparameter_optimization(
        eval_fn=eval_fn, hyperparams=params, search_key_ranges=ranges, n=n, save_results_to=save_results_to, m=m,
        metric_should_increase=metric_should_increase, metric_name=metric_name)
    # (the reference indexes best_config[lr_name] and raises KeyError when the lr range is a single point; the fixed value is returned here)
    lr = best_config[lr_name] if lr_name in best_config else ranges[lr_name][0]
    return float(lr), int(expected_best_step_num)
