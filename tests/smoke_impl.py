"""smoke(): one small inner step of the hot path on cuda:0, checked against the CPU oracle."""
import torch


def run_smoke():
    from oracle import efficientlab_ref as R
    from mliis_amd.learner import Learner
    from mliis_amd.hostinfo import usable_cores
    from mliis_amd.metaseg import synthetic_task
    torch.set_num_threads(usable_cores())     # (the oracle step: the box shows more CPUs than its cgroup quota grants)
    H, S, idx = 64, 5, [0, 1, 2, 3, 4, 0, 1, 2]
    O = R.OracleLearner(image_size=H, seed=0, dtype=torch.float64)
    L = Learner(image_size=H, seed=1, use_graph=False, drop_connect=False)
    L.load_named({k: v.numpy() for k, v in O.params.items()}, strict=False)
    x, y = synthetic_task(S, H, seed=0)
    L.load_task(x, y)
    lo = O.inner_step(torch.tensor(x[idx]).double(), torch.tensor(y[idx]).double())
    L.inner_step(idx)
    ll = L.loss_value()
    assert abs(ll - lo) <= 1e-4 * max(1.0, abs(lo)), (ll, lo)
    th = L.arena.export_trainable_packed().cpu().double()
    ref = torch.cat([O.params[p.name].reshape(-1) for p in L.arena.trainable])
    assert (th - ref).abs().max().item() <= 1e-5
    print("smoke ok: loss hip %.6f oracle %.6f" % (ll, lo))
