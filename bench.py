"""Headline benchmark: inner-loop images/sec of the Reptile/FOMAML adaptation loop over EfficientLab-6-3 (224x224, 5-shot).

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one meta-step: every rank adapts ONE synthetic 5-shot task (BASELINE.json configs[1]: 8 inner SGD steps of batch 8
= 64 image passes, fp32, drop-connect on), then the outer Reptile update (one RCCL all-reduce of the flat delta when N > 1).
Meta-batch = N tasks, one per GPU (weak scaling, no data-path collective besides that exchange).  Synthetic data of
BASELINE.md section 3, resident in HBM before the timed region.  Rank 0 prints ONE JSON line with `roofline` (dominant kernel,
timed live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); concurrent task lanes (--concurrent-tasks)
# want one queue each next to torch's own streams.  Read when the HIP runtime loads, so set before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, dense f32-input MFMA
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA (the --precision bf16 variant only)
HBM_PEAK_GBS = 8000.0          # ibid., HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--foml", action="store_true", help="FOMAML with a 5-shot tail batch (BASELINE configs[2] flavour)")
    ap.add_argument("--inner-iters", type=int, default=8)
    ap.add_argument("--inner-batch", type=int, default=8)
    ap.add_argument("--shots", type=int, default=5)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--backbone", default="efficientnet-b0", choices=["efficientnet-b0", "efficientnet-b3"],
                    help="variant: EfficientNet-B3 encoder (BASELINE configs[3]); the metric's config is B0")
    ap.add_argument("--aspp", action="store_true", help="variant: ASPP decoder in front of the RSD modules (--spatial_pyramid_pooling)")
    ap.add_argument("--augment", action="store_true", help="variant: host augmentation of every inner-loop batch (the reference's run.sh setting)")
    ap.add_argument("--augment-workers", type=int, default=-1, help="worker processes for the augmentation pixel work (0 = inline, -1 = cores - 1)")
    ap.add_argument("--overlap-wgrad", type=int, default=0, help="variant: weight-gradient kernels on a second stream inside the graph (1 | 2)")
    ap.add_argument("--concurrent-tasks", type=int, default=1,
                    help="variant (with --tasks-per-gpu > 1): adapt this many tasks of the meta-batch at once on separate learners / streams")
    ap.add_argument("--tasks-per-gpu", type=int, default=1, help="variant: tasks per GPU and meta-step (the metric's config has 1)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="variant: bf16 operands on the matrix cores (fp32 accumulation, fp32 tensors); the headline metric is fp32")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--pool", type=int, default=8, help="number of distinct synthetic tasks resident per GPU")
    return ap.parse_args()


def usable_cores() -> int:
    """Host cores this process may actually use: affinity mask, further limited by a cgroup CPU quota if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(args):
    """CPU oracle (PyTorch-CPU fp32 restatement of the same graph) on a bounded sample of the same workload."""
    from oracle import efficientlab_ref as R
    from mliis_amd.metaseg import synthetic_task, mini_batch_indices
    import random
    cores = usable_cores()
    torch.set_num_threads(cores)
    O = R.OracleLearner(name=args.backbone, image_size=args.image_size, seed=0, dtype=torch.float32, lr=1e-3, aspp=args.aspp)
    x, y = synthetic_task(args.shots, args.image_size, seed=0)
    O.load_task(torch.tensor(x), torch.tensor(y))
    batches = [list(b) for b in mini_batch_indices(args.shots, args.inner_batch, 64, rng=random.Random(0))]
    t0 = time.time()
    O.inner_step(batches[0])           # warm-up step (also a cost probe)
    probe = time.time() - t0
    n = int(max(2, min(args.inner_iters, 20.0 / max(probe, 1e-3))))
    t0 = time.time()
    for b in batches[1:1 + n]:
        O.inner_step(b)
    dt = time.time() - t0
    return {"value": n * args.inner_batch / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d inner SGD steps (batch %d, %dx%d, fp32) of one synthetic %d-shot task on the PyTorch-CPU oracle, "
                      "%d threads, after 1 warm-up step" % (n, args.inner_batch, args.image_size, args.image_size, args.shots, cores)}


def roofline(L, args):
    """Per-launch timing (HIP events on the learner's stream) of one eager inner step; reports the dominant kernel."""
    from mliis_amd import ops
    idx = [i % args.shots for i in range(args.inner_batch)]
    reps = 5
    L.use_graph, saved = False, L.use_graph
    L.inner_step(idx)
    ops.PROFILE = []
    for _ in range(reps):
        L.inner_step(idx)
    L.synchronize()
    recs = ops.profile_resolve(ops.PROFILE)
    ops.PROFILE = None
    L.use_graph = saved
    by = {}
    for r in recs:
        if r["op"] in ("conv2d_fwd", "conv2d_bwd_data") and r.get("splits", 1) == 1:
            k = r["kernel"]      # exact instantiation name; split-K launches are a different instantiation (their op time includes the fold)
        elif r["op"] == "conv2d_bwd_filter":
            k = "conv_filter_grad_k(+reduce)"
        elif r["op"].startswith("dwconv"):
            k = r["op"]
        else:
            continue
        d = by.setdefault(k, dict(ms=0.0, n=0, flops=0.0, bytes=0.0))
        d["ms"] += r["ms"]
        d["n"] += 1
        d["flops"] += r.get("flops", 0.0)
        d["bytes"] += r.get("bytes", 0.0)
    gemm = {k: v for k, v in by.items() if k.startswith("conv_gemm")}
    dom = max(gemm, key=lambda k: gemm[k]["ms"])
    d = gemm[dom]
    # The per-op event pairs above include a few microseconds of dispatch per launch.  For the dominant kernel re-issue every one
    # of its launches of an inner step `burst` times back to back between two HIP events on the learner's stream, so that the
    # figure is the kernel's own duration (what rocprofv3 --kernel-trace reports, profiles/r01_final_kernel_stats.csv).
    import torch

    def burst_ms(sites, burst=20):
        total = 0.0
        with torch.cuda.stream(L.stream):
            for r in sites:
                r["fn"]()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(L.stream)
                for _ in range(burst):
                    r["fn"]()
                e1.record(L.stream)
                e1.synchronize()
                total += e0.elapsed_time(e1) / burst
        return total
    per_step = d["n"] // reps
    sites = [r for r in recs if r.get("kernel") == dom and r["op"] in ("conv2d_fwd", "conv2d_bwd_data")][-per_step:]
    ms, fl = burst_ms(sites), sum(r["flops"] for r in sites)
    d = dict(d, ms=ms, n=len(sites), flops=fl)
    reps_dom = 1
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    traffic, tsrc = None, None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath):   # PMC counters cannot be sampled from inside this process: taken from the committed rocprofv3 --pmc passes
        k = json.load(open(tpath)).get("kernels", {}).get(dom.replace(", ", ","))
        if k:
            traffic, tsrc = k["hbm_bytes_per_launch"], "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected)"
    peak = MFMA_F32_PEAK_TFLOPS if args.precision == "fp32" else MFMA_BF16_PEAK_TFLOPS
    out = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
           "traffic": traffic, "traffic_source": tsrc, "launches_per_step": d["n"] // reps_dom, "avg_launch_us": 1e3 * d["ms"] / d["n"],
           "algorithmic_flops_per_launch": d["flops"] / d["n"]}
    # depthwise families against the HBM roofline: algorithmic bytes / the kernels' own durations.  The launches of one
    # eager step are recorded at the C ABI and re-issued back to back straight through ctypes (a Python-level wrapper call costs
    # about as much as one of the 5 us kernels and would be what is timed).
    from mliis_amd._lib import lib as _lib
    from mliis_amd.spec import same_pad
    L.use_graph = False
    _lib.trace = []
    L.inner_step(idx)
    L.synchronize()
    calls, _lib.trace = _lib.trace, None
    L.use_graph = saved
    dll = _lib.load()
    dw = {}
    for k in ("dwconv_fwd", "dwconv_bwd_data", "dwconv_bwd_filter"):
        ksites = [(n, a) for n, a in calls if n == "mliis_" + k or (k == "dwconv_bwd_data" and n == "mliis_dwconv_bwd_data_bn")]
        if not ksites:
            continue
        kms, kbytes = 0.0, 0.0
        with torch.cuda.stream(L.stream):
            for n, a in ksites:
                fn = getattr(dll, n)
                fn(*a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(L.stream)
                for _ in range(20):
                    fn(*a)
                e1.record(L.stream)
                e1.synchronize()
                kms += e0.elapsed_time(e1) / 20
                nb, h, w_, c, kk, st = a[3:9]
                ho, wo = same_pad(h, kk, st)[0], same_pad(w_, kk, st)[0]
                i_el, o_el, w_el = nb * h * w_ * c, nb * ho * wo * c, kk * kk * c
                kbytes += 4.0 * (i_el + o_el + w_el)   # fwd: X + Y + W; bwd-data: dY + dX + W; bwd-filter: X + dY + dW
                if n.endswith("_bn"):
                    kbytes += 4.0 * i_el               # ... + z0: the launch also produces the expand BN's backward statistics
        dw[k] = {"us_per_step": 1e3 * kms, "launches_per_step": len(ksites), "algorithmic_MB_per_step": kbytes / 1e6,
                 "GBps": kbytes / (kms * 1e-3) / 1e9, "frac_of_8TBps": kbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    families = {k: {"us_per_step": 1e3 * v["ms"] / reps, "launches_per_step": v["n"] // reps,
                    "TFLOPs": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] else None} for k, v in sorted(by.items())}
    return out, dw, families


def main():
    import contextlib
    args = parse()
    with contextlib.redirect_stdout(sys.stderr):   # stdout carries exactly ONE JSON line
        out = _run(args)
    if out is not None:
        print(json.dumps(out))


def _run(args):
    aug_pool = None
    if args.augment and args.augment_workers != 0:   # forked workers: before anything initialises the GPU
        from mliis_amd.augment import AugmentPool
        aug_pool = AugmentPool(None if args.augment_workers < 0 else args.augment_workers)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path (the CPU oracle is only the reported baseline)")
    device = torch.device("cuda", local)
    from mliis_amd.learner import Learner
    from mliis_amd.metaseg import DeviceTask, synthetic_task
    from mliis_amd.reptile import FOMLIS, Gecko, Dist

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    shots = 10 if args.foml else args.shots
    L = Learner(feature_extractor_name=args.backbone, image_size=args.image_size, rsd=(2, 4), learning_rate=1e-3, optimizer="sgd", dice=False, l2=False, seed=0, device=device,
                use_graph=not args.no_graph, max_shots=max(16, shots), spatial_pyramid_pooling=args.aspp, matmul_precision=args.precision,
                overlap_wgrad=args.overlap_wgrad)
    lanes = [Learner(feature_extractor_name=args.backbone, image_size=args.image_size, rsd=(2, 4), learning_rate=1e-3, optimizer="sgd", dice=False,
                     l2=False, seed=k, device=device, use_graph=not args.no_graph, max_shots=max(16, shots),
                     spatial_pyramid_pooling=args.aspp, matmul_precision=args.precision) for k in range(1, args.concurrent_tasks)]
    tasks = []
    for i in range(args.pool):
        x, y = synthetic_task(shots, args.image_size, seed=1000 * rank + i)
        tasks.append(DeviceTask("synthetic_%d_%d" % (rank, i), torch.from_numpy(x).to(device), torch.from_numpy(y).to(device)))
    D = Dist()
    if args.foml:
        meta = FOMLIS(L, train_shots=shots, tail_shots=5, dist=D, rng_mode="per_task", seed=0, augment=args.augment, aug_rate=0.5, aug_pool=aug_pool,
                      lanes=lanes)
    else:
        meta = Gecko(L, dist=D, rng_mode="per_task", seed=0, augment=args.augment, aug_rate=0.5, aug_pool=aug_pool, lanes=lanes)

    def step():
        meta.train_step(tasks, num_shots=shots, inner_batch_size=args.inner_batch, inner_iters=args.inner_iters, replacement=False,
                        meta_step_size=0.1, meta_batch_size=world * args.tasks_per_gpu)

    for _ in range(args.warmup):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    D.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    imgs_per_task = ((args.inner_iters - 1) * args.inner_batch + 5) if args.foml else args.inner_iters * args.inner_batch
    value = world * args.tasks_per_gpu * imgs_per_task * args.steps / dt
    loss = L.loss_value()

    roof = dwr = fam = None
    if rank == 0 and not args.no_roofline:
        roof, dwr, fam = roofline(L, args)
    if rank == 0:
        out = {
            "metric": "inner-loop images/sec (EfficientLab-6-3, 224x224, 5-shot)", "value": value, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16 matrix-core operands, f32 accumulate / tensors", "data": "synthetic",
            "config": {"workload": "%s + %sRSD(4)+RSD(2)), %dx%d, meta-batch=%d (1 task/GPU), "
                                   "%d-shot x %d inner SGD steps of batch %d (%d image fwd+bwd+BN-EMA+SGD per task), %s outer update, fp32 tensors, "
                                   "CE loss, drop-connect on%s" % ("EfficientLab-6-3 (EfficientNet-B0 blocks 0-10" if args.backbone == "efficientnet-b0" else
                                                                 "EfficientLab with the EfficientNet-B3 encoder (blocks 0-17",
                                                                 "ASPP + " if args.aspp else "", args.image_size, args.image_size, world, shots,
                                                                 args.inner_iters,
                                                                 args.inner_batch, imgs_per_task, "FOMAML(tail 5)" if args.foml else "Reptile",
                                                                 ("" if args.precision == "fp32" else ", bf16 matrix-core operands") +
                                                                 ((", host augmentation (aug_rate 0.5, %s)" % ("%d worker processes" % aug_pool.workers if aug_pool else "inline"))
                                                                  if args.augment else "") +
                                                                 (", %d tasks per GPU and meta-step" % args.tasks_per_gpu if args.tasks_per_gpu != 1 else "") +
                                                                 (", %d adapted concurrently" % args.concurrent_tasks if args.concurrent_tasks != 1 else "")),
                       "hip_graph": not args.no_graph, "final_loss": loss},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if dwr is not None:
            out["depthwise_hbm"] = dwr
            out["kernel_families_eager_us"] = fam
        if cpu is not None:
            out["gpu_over_cpu"] = value / cpu["value"]
    else:
        out = None
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                      # rank 0 may still be in its roofline pass
        dist.destroy_process_group()
    if aug_pool is not None:
        aug_pool.close()
    return out


if __name__ == "__main__":
    main()
