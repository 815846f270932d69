"""Headline benchmark: inner-loop images/sec of the Reptile/FOMAML adaptation loop over EfficientLab-6-3 (224x224, 5-shot).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Launch contract: for N > 1 the launcher (torch.distributed.run) exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; this process
binds cuda:LOCAL_RANK and joins the nccl (= RCCL) group BEFORE its first GPU call.  A plain `python bench.py --gpus N` (N > 1, no
WORLD_SIZE in the environment) starts that launcher ITSELF as a child process -- before anything has touched the GPU, nothing is
re-executed in place -- relays rank 0's JSON line and exits with the child's code.  A WORLD_SIZE that disagrees with --gpus is an
error (exit 2): the line's `n_gpus` is always the number of ranks that really ran.

One "step" = one meta-step: every rank adapts ONE synthetic 5-shot task (BASELINE.json configs[1]: 8 inner SGD steps of batch 8
= 64 image passes, fp32, drop-connect on), then the outer Reptile update (one RCCL all-reduce of the flat delta when N > 1).
Meta-batch = N tasks, one per GPU (weak scaling, no data-path collective besides that exchange).  Synthetic data of
BASELINE.md section 3, resident in HBM before the timed region.  Rank 0 prints ONE JSON line with `roofline` (dominant kernel,
timed live with HIP events on the launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores).
"""
import argparse
import json
import os
import re
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


def parse(argv=None):
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
    ap.add_argument("--skip-decoding", action="store_true", help="variant: DeepLabv3+-style decoder in front of the RSD modules (--skip_decoding)")
    ap.add_argument("--augment", action="store_true", help="variant: augmentation of every inner-loop batch (the reference's run.sh setting), pixels on the device")
    ap.add_argument("--augment-on-host", action="store_true", help="with --augment: pixels in numpy / scipy on the host (draw-identical to the reference)")
    ap.add_argument("--augment-workers", type=int, default=-1, help="worker processes for the augmentation pixel work (0 = inline, -1 = cores - 1)")
    ap.add_argument("--concurrent-tasks", type=int, default=1,
                    help="variant (with --tasks-per-gpu > 1): adapt this many tasks of the meta-batch at once on separate learners / streams")
    ap.add_argument("--tasks-per-gpu", type=int, default=1, help="variant: tasks per GPU and meta-step (the metric's config has 1)")
    ap.add_argument("--precision", choices=["fp32", "fp32-native", "bf16", "fp8", "bf16-storage"], default="fp32",
                    help="variant: bf16 / fp8 operands on the matrix cores (fp32 accumulation, fp32 tensors); bf16-storage: bf16 operands AND the "
                         "expanded MBConv tensors as bf16 in HBM (BASELINE configs[3]); the headline metric is fp32")
    ap.add_argument("--adam", action="store_true", help="variant: Adam(beta1 = 0) inner optimizer, the reference's default when --sgd is absent (the metric's config is SGD)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-native-retime", action="store_true", help="skip the extra `fp32_native_images_per_s` key (a second learner on the fp32 matrix instruction, 5 tasks)")
    ap.add_argument("--pool", type=int, default=8, help="number of distinct synthetic tasks resident per GPU")
    ap.add_argument("--cpu-baseline-from", default=None,
                    help="N > 1 lines do not time the CPU oracle; carry `cpu_baseline` by value from this N = 1 bench line (a JSON file, e.g. BENCH_rNN.json)")
    return ap.parse_args(argv)


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_baseline_fn(args):   # (a seam for tests/test_bench_cli_cpu.py)
    return cpu_baseline(args)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 outside a launcher: run torch.distributed.run as a CHILD (one rank per GPU), relay its
    stdout (rank 0's one JSON line; the ranks' stderr passes through) and return its exit code.  Called before any GPU call."""
    import subprocess
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    # the CPU baseline of an N > 1 line is timed HERE, in the parent (which never touches the GPU), before the ranks start, and handed
    # to rank 0 by value: a SCALE line carries `cpu_baseline` without --cpu-baseline-from
    if not args.no_cpu_baseline and not args.cpu_baseline_from and "MLIIS_BENCH_CPU_BASELINE" not in env:
        env["MLIIS_BENCH_CPU_BASELINE"] = json.dumps(dict(_cpu_baseline_fn(args), timed_in="the launching parent, before the ranks started"))
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    text = p.stdout.decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.lstrip().startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    for ln in text.splitlines():
        if not ln.lstrip().startswith("{"):
            print(ln, file=sys.stderr)
    if p.returncode == 0 and not lines:
        print("bench.py: the %d-rank child printed no JSON line" % args.gpus, file=sys.stderr)
        return 3
    return p.returncode


def check_world(args, environ=None):
    """--gpus against the launcher's WORLD_SIZE.  Returns "launch" (spawn the ranks), "run" (this process is a rank / the only rank)."""
    environ = os.environ if environ is None else environ
    ws = environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if ws is None:
        return "launch" if args.gpus > 1 else "run"
    if int(ws) != args.gpus:
        print("bench.py: --gpus %d but the launcher's WORLD_SIZE is %s; refusing to report a line whose n_gpus differs from the request"
              % (args.gpus, ws), file=sys.stderr)
        raise SystemExit(2)
    return "run"


def _rccl_version():
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:   # noqa: BLE001  (reporting only)
        return None


from mliis_amd.hostinfo import usable_cores  # noqa: E402


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
    # at least 3 warm-up steps, then at least 2 full tasks (2 x inner_iters steps), bounded to ~25 s of CPU work
    WARM = 3
    batches = [list(b) for b in mini_batch_indices(args.shots, args.inner_batch, WARM + 4 * args.inner_iters, rng=random.Random(0))]
    t0 = time.time()
    for b in batches[:WARM]:
        O.inner_step(b)
    probe = (time.time() - t0) / WARM
    n = 2 * args.inner_iters
    if n * probe > 25.0:   # (a slow box: as many whole steps as fit, never fewer than one task)
        n = int(max(args.inner_iters, 25.0 / max(probe, 1e-3)))
    t0 = time.time()
    for b in batches[WARM:WARM + n]:
        O.inner_step(b)
    dt = time.time() - t0
    return {"value": n * args.inner_batch / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d inner SGD steps = %.2g tasks (batch %d, %dx%d, fp32) of synthetic %d-shot data on the PyTorch-CPU oracle, "
                      "%d threads, after %d warm-up steps" % (n, n / float(args.inner_iters), args.inner_batch, args.image_size, args.image_size,
                                                              args.shots, cores, WARM)}


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
        if r["op"] in ("conv2d_fwd", "conv2d_bwd_data", "conv2d_fwd_x3", "conv2d_bwd_data_x3") and r.get("splits", 1) == 1:
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
    gemm = {k: v for k, v in by.items() if k.startswith(("conv_gemm", "conv_x3"))}
    dom = max(gemm, key=lambda k: gemm[k]["ms"])
    d = gemm[dom]
    # The per-op event pairs above include a few microseconds of dispatch per launch.  For the dominant kernel re-issue every one
    # of its launches of an inner step `burst` times back to back between two HIP events on the learner's stream, so that the
    # figure is the kernel's own duration (what rocprofv3 --kernel-trace reports, profiles/r02_final_kernel_stats.csv).
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
    sites = [r for r in recs if r.get("kernel") == dom and r["op"] in ("conv2d_fwd", "conv2d_bwd_data", "conv2d_fwd_x3", "conv2d_bwd_data_x3")][-per_step:]
    ms, fl = burst_ms(sites), sum(r["flops"] for r in sites)
    d = dict(d, ms=ms, n=len(sites), flops=fl)
    reps_dom = 1
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    traffic, tsrc = None, None
    # newest committed rocprofv3 --pmc summary (profiles/rNN_pmc_traffic.json, tools/final_profile.sh)
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.match(r"r\d+_pmc_traffic\.json$", f))
    tpath = os.path.join(ROOT, "profiles", cands[-1]) if cands else ""
    if os.path.exists(tpath):   # PMC counters cannot be sampled from inside this process: taken from the committed rocprofv3 --pmc passes
        tj = json.load(open(tpath))
        k = tj.get("kernels", {}).get(dom.replace(", ", ","))
        if k:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from pmc_traffic import csrc_sha1
            current = tj.get("csrc_sha1") == csrc_sha1()     # collected on exactly the kernel sources this run was built from?
            traffic = k["hbm_bytes_per_launch"]
            tsrc = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950-corrected; %s)" % (
                os.path.basename(tpath), "collected on the current kernel sources" if current else "STALE: the kernel sources changed since it was collected")
    x3 = dom.startswith("conv_x3")
    # the split-product kernels (fp32 operands as three bf16 terms, SIX bf16 term products per fp32 product): `achieved` counts the fp32
    # products (2 M K N), so the peak that bounds them is the dense bf16 matrix rate / 6; the fp32 instruction's own peak is beside it
    peak = MFMA_BF16_PEAK_TFLOPS / 6.0 if x3 else (MFMA_F32_PEAK_TFLOPS if args.precision in ("fp32", "fp32-native") else MFMA_BF16_PEAK_TFLOPS)
    out = {"bound": "mfma", "kernel": dom + (" (+ x3_fixup_k: the stream-K fix-up launch is inside the timed region)" if x3 else ""),
           "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
           "peak_note": ("dense bf16 MFMA peak 2500 TFLOP/s / 6 term products per fp32 product; the fp32 MFMA instruction's peak is %.1f "
                         "TFLOP/s (achieved / that = %.2f)" % (MFMA_F32_PEAK_TFLOPS, ach / MFMA_F32_PEAK_TFLOPS)) if x3 else None,
           "traffic": traffic, "traffic_source": tsrc, "launches_per_step": d["n"] // reps_dom, "avg_launch_us": 1e3 * d["ms"] / d["n"],
           "algorithmic_flops_per_launch": d["flops"] / d["n"]}
    # the whole inner step against both rooflines (filled in by _run, which knows the measured step time)
    out["_issued_flops_per_step"] = sum(r.get("flops", 0.0) for r in recs) / reps
    out["_hbm_bytes_per_step"] = (tj.get("hbm_bytes_per_inner_step") if os.path.exists(tpath) else None)
    # depthwise families against the HBM roofline: algorithmic bytes / the kernels' own durations, measured COLD: the
    # launches of one eager step are recorded at the C ABI and re-issued straight through ctypes (a Python-level wrapper call costs
    # about as much as one of these kernels) over ROTATING copies of their activation operands -- enough copies that a tensor is
    # only touched again after > 320 MB of other traffic, i.e. after the 256 MiB Infinity Cache has been flushed.  (Re-issuing one
    # launch back to back on the same <= 39 MB tensors, as round 1 did, times the cache, not HBM.)  Inside a real step the operands are
    # partly cache-resident (the producer ran just before), so the in-step durations of profiles/r02_*_kernel_stats.csv are shorter.
    from mliis_amd._lib import lib as _lib
    from mliis_amd.spec import same_pad
    L.use_graph = False
    _lib.trace = []
    L.inner_step(idx)
    L.synchronize()
    calls, _lib.trace = _lib.trace, None
    L.use_graph = saved
    dll = _lib.load()
    # entry point -> (family, index of N in the argument list, {pointer argument: "i" input-resolution | "o" output-resolution tensor},
    #                 pointer arguments nulled in the re-issues (moving averages: the learner's own state must not be advanced))
    # (the row-marching kernels: forward = bn0 fold + apply + swish while staging + depthwise conv + bn1 stage-1 sums; backward = ONE pass
    #  over (dy, z): dx, filter-gradient slabs, stage 1 of bn0's backward.  Their mean / rstd outputs go to scratch in the re-issues)
    DW = {"mliis_dwconv_bn_fwd": ("dwconv_bn_fwd", 13, {0: "i", 12: "o"}, (7, 8)),
          "mliis_dwconv_bn_bwd": ("dwconv_bn_bwd", 9, {0: "o", 1: "i", 7: "i"}, ()),
          # (the same pass with the depthwise batch norm's backward apply formed while (da2, z1) are staged: it ALSO reads z1, 4 * out
          #  bytes that the depthwise formula below does not count -- its fraction is understated by that much)
          "mliis_mbconv_dw_bwd_march": ("dwconv_bn_bwd", 19, {0: "o", 1: "o", 12: "i", 18: "i"}, ()),
          "mliis_dwconv_fwd": ("dwconv_fwd", 3, {0: "i", 2: "o"}, ()),
          "mliis_dwconv_bwd_data": ("dwconv_bwd_data", 3, {0: "o", 2: "i"}, ()),
          "mliis_dwconv_bwd_data_bn": ("dwconv_bwd_data", 3, {0: "o", 2: "i", 9: "i"}, ()),
          "mliis_dwconv_bwd_filter": ("dwconv_bwd_filter", 3, {0: "i", 1: "o"}, ()),
          # (+ the group-blocked copy of z0 the forward leaves for the backward: written / read, so it rotates with the other tensors)
          "mliis_mbconv_dw_fwd_small": ("mbconv_small_fwd", 20, {0: "i", 17: "o", 18: "o", 29: "i"}, (7, 8, 14, 15)),
          "mliis_mbconv_dw_bwd_small": ("mbconv_small_bwd", 20, {0: "o", 3: "o", 9: "i", 19: "i", 27: "i"}, ())}
    FLUSH = 320e6
    dw, layers = {}, []
    with torch.cuda.stream(L.stream):
        for n, a in calls:
            if n not in DW:
                continue
            fam_name, iN, rot, nulled = DW[n]
            nb, h, w_, c, kk = a[iN:iN + 5]
            st = a[iN + 5] if fam_name.startswith("dwconv") else 1
            ho, wo = same_pad(h, kk, st)[0], same_pad(w_, kk, st)[0]
            el = {"i": nb * h * w_ * c, "o": nb * ho * wo * c}
            w_el = kk * kk * c
            if fam_name in ("mbconv_small_bwd", "dwconv_bn_bwd"):   # backward-data + backward-filter of the depthwise op: dY, X read, dX written, W, dW
                nbytes = 4.0 * (2 * el["i"] + el["o"] + 2 * w_el)
            else:                                # fwd: X + Y + W; bwd-data: dY + dX + W (+ z0 when the BN sums ride along); bwd-filter: X + dY + dW
                nbytes = 4.0 * (el["i"] + el["o"] + w_el) + (4.0 * el["i"] if n.endswith("_bn") else 0.0)
            copies = max(2, int(FLUSH / nbytes) + 1)
            bufs, variants = [], []
            for r in range(copies):
                args_r = list(a)
                same = {}                       # arguments that ALIAS in the recorded call alias in the re-issue too (a layer whose producer
                for ix, which in rot.items():   # already wrote the group-blocked z0 passes one tensor as z0 and as its blocked copy)
                    if args_r[ix] is None:      # (a nullable tensor argument that this launch does not use)
                        continue
                    if a[ix] in same:
                        args_r[ix] = same[a[ix]]
                        continue
                    buf = torch.empty(int(el[which]), dtype=torch.float32, device=L.device).normal_()
                    bufs.append(buf)
                    args_r[ix] = same[a[ix]] = buf.data_ptr()
                for ix in nulled:
                    args_r[ix] = None
                if n == "mliis_dwconv_bn_fwd" and args_r[3] is not None:   # batch statistics the launch writes: not into the learner's
                    scr = torch.empty(2 * c, dtype=torch.float32, device=L.device)
                    bufs.append(scr)
                    args_r[5], args_r[6] = scr.data_ptr(), scr[c:].data_ptr()
                variants.append(tuple(args_r))
            fn = getattr(dll, n)
            for v in variants:          # first touch of every copy (page mapping) outside the timed region
                fn(*v)
            reps_k = max(20, 2 * copies)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(L.stream)
            for r in range(reps_k):
                fn(*variants[r % copies])
            e1.record(L.stream)
            e1.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps_k
            del variants, bufs
            # practical ceiling at THIS size: a plain device copy that moves the same number of bytes (half read, half written), timed the
            # same way (cold, rotating) -- a 25 MB launch cannot reach the 8 TB/s of a long stream: ramp-up, first-touch latency and drain
            # are a fixed ~3-4 us
            half = int(nbytes // 8)
            csrc = [torch.empty(half, dtype=torch.float32, device=L.device).normal_() for _ in range(copies)]
            cdst = [torch.empty(half, dtype=torch.float32, device=L.device) for _ in range(copies)]
            for a_, b_ in zip(csrc, cdst):
                b_.copy_(a_)
            cus_ = []
            for _ in range(3):   # (median of three bursts: a single burst of a 4-8 us copy is noisy -- VERDICT r02)
                e0.record(L.stream)
                for r in range(reps_k):
                    cdst[r % copies].copy_(csrc[r % copies])
                e1.record(L.stream)
                e1.synchronize()
                cus_.append(1e3 * e0.elapsed_time(e1) / reps_k)
            cus = sorted(cus_)[1]
            del csrc, cdst
            layers.append({"entry": n[6:], "N,H,W,C,k,s": [nb, h, w_, c, kk, st], "us": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 2),
                           "frac_of_8TBps": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 3), "rotating_copies": copies,
                           "copy_same_bytes_us": round(cus, 2), "frac_of_copy": round(cus / us, 3)})
            d_copy = dw.setdefault("_copy_us", {})
            d_copy[fam_name] = d_copy.get(fam_name, 0.0) + cus
            d_ = dw.setdefault(fam_name, {"us_per_step": 0.0, "launches_per_step": 0, "algorithmic_MB_per_step": 0.0})
            d_["us_per_step"] += us
            d_["launches_per_step"] += 1
            d_["algorithmic_MB_per_step"] += nbytes / 1e6
    copy_us = dw.pop("_copy_us", {})
    for k_, d_ in dw.items():
        d_["GBps"] = d_["algorithmic_MB_per_step"] * 1e6 / (d_["us_per_step"] * 1e-6) / 1e9
        d_["frac_of_8TBps"] = d_["GBps"] / HBM_PEAK_GBS
        d_["copy_same_bytes_us_per_step"] = copy_us.get(k_)
        d_["frac_of_copy"] = (copy_us[k_] / d_["us_per_step"]) if k_ in copy_us else None
    tot_us = sum(d_["us_per_step"] for d_ in dw.values())
    tot_mb = sum(d_["algorithmic_MB_per_step"] for d_ in dw.values())
    dw["all_depthwise_fwd_bwd"] = {"us_per_step": tot_us, "algorithmic_MB_per_step": tot_mb, "frac_of_8TBps": tot_mb * 1e6 / (tot_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                   "copy_same_bytes_us_per_step": sum(copy_us.values()), "frac_of_copy": sum(copy_us.values()) / tot_us,
                                   "method": "cold operands: rotating copies, > 320 MB between two touches of a tensor (Infinity Cache flushed)"}
    dw["per_layer"] = layers
    bnr = batchnorm_hbm(L, calls, dll, FLUSH) if args.precision in ("fp32", "fp32-native") else None
    families = {k: {"us_per_step": 1e3 * v["ms"] / reps, "launches_per_step": v["n"] // reps,
                    "TFLOPs": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["flops"] else None} for k, v in sorted(by.items())}
    return out, dw, families, bnr


def retime_native(args, device, tasks, shots, imgs_per_task):
    """The same workload on a second learner that uses the native fp32 matrix instruction everywhere (`--precision fp32-native`), timed
    after the headline region: 2 warm-up tasks + 5 timed tasks on this rank alone (no collective)."""
    import torch
    from mliis_amd.learner import Learner
    from mliis_amd.reptile import FOMLIS, Gecko, SingleRank
    L2 = Learner(feature_extractor_name=args.backbone, image_size=args.image_size, rsd=(2, 4), learning_rate=1e-3, optimizer="adam" if args.adam else "sgd", dice=False,
                 l2=False, seed=0, device=device, use_graph=not args.no_graph, max_shots=max(16, shots), spatial_pyramid_pooling=args.aspp,
                 matmul_precision="fp32-native", skip_decoding=args.skip_decoding)
    kw = dict(dist=SingleRank(), rng_mode="per_task", seed=0)
    meta = FOMLIS(L2, train_shots=shots, tail_shots=5, **kw) if args.foml else Gecko(L2, **kw)

    def step():
        meta.train_step(tasks, num_shots=shots, inner_batch_size=args.inner_batch, inner_iters=args.inner_iters, replacement=False, meta_step_size=0.1,
                        meta_batch_size=1)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L2.close()
    return {"value": n * imgs_per_task / dt, "note": "same workload, every matrix product on v_mfma_f32_16x16x4_f32 (--precision fp32-native): 2 warm-up + %d timed tasks on "
                                                    "rank 0 after the headline region" % n}


def batchnorm_hbm(L, calls, dll, flush_bytes):
    """The batch-norm launches of one inner step against the HBM roofline, measured like `depthwise_hbm`: every recorded C-ABI call of
    the family is re-issued through ctypes over ROTATING copies of its activation tensors (> 320 MB between two touches of a tensor:
    the Infinity Cache is flushed), with a device copy of the same number of bytes timed the same way beside it.  Algorithmic bytes:
    every activation tensor of the call once (x read, y written, residual read; backward: x, dy read, dx written, the skip gradient
    written or read + written) -- the backward launches that form their own sums first (no stage 1 from the producer) read x and dy
    TWICE by design; that second pass is NOT counted as algorithmic, it shows as a lower fraction (fp32 tensors only)."""
    import torch
    # entry -> (family, rows index, C index, [(pointer index, ld index or None, role)], indices nulled, indices redirected to scratch of C floats)
    BN = {
        "mliis_bn_apply_fused": ("bn_apply_fused", 4, 5, [(0, 1, "r"), (2, 3, "w"), (21, 22, "r")], (14, 15), (12, 13)),
        "mliis_bn_apply_fused_pair": ("bn_apply_fused", 22, 23, [(0, 20, "r"), (1, 21, "w"), (10, 20, "r"), (11, 21, "w")], (6, 7, 16, 17), (4, 5, 14, 15)),
        "mliis_bn_bwd": ("bn_bwd", 6, 7, [(0, 1, "r"), (2, 3, "r"), (4, 5, "w"), (20, 21, "s")], (), ()),
        "mliis_bn_bwd_pair": ("bn_bwd", 23, 24, [(0, 20, "r"), (1, 21, "r"), (2, 22, "w"), (10, 20, "r"), (11, 21, "r"), (12, 22, "w")], (), ()),
        "mliis_bn_stats_partial": ("bn_stats", 2, 3, [(0, 1, "r")], (), ()),
    }
    fams, layers = {}, []
    with torch.cuda.stream(L.stream):
        for n, a in calls:
            if n not in BN:
                continue
            fam, i_rows, i_c, tensors, nulled, scratch = BN[n]
            if n in ("mliis_bn_apply_fused", "mliis_bn_bwd") and a[26 if n == "mliis_bn_apply_fused" else 29] != 0:
                continue                                    # (bf16 tensors: not this table)
            rows, c = int(a[i_rows]), int(a[i_c])
            nbytes, live = 0.0, []
            for ip, ild, role in tensors:
                if a[ip] is None or a[ip] == 0:
                    continue
                ld = int(a[ild]) if ild is not None else c
                acc = role == "s" and n == "mliis_bn_bwd" and int(a[22]) != 0     # skip gradient accumulated: read + written
                nbytes += 4.0 * rows * c * (2 if acc else 1)
                live.append((ip, rows * ld))
            two_pass = (n == "mliis_bn_bwd" and (a[27] is None or a[27] == 0)) or n == "mliis_bn_bwd_pair"
            copies = max(2, int(flush_bytes / nbytes) + 1)
            bufs, variants = [], []
            for r in range(copies):
                args_r, same = list(a), {}
                for ip, nel in live:
                    if a[ip] in same:                       # (operands that alias in the recorded call alias in the re-issue)
                        args_r[ip] = same[a[ip]]
                        continue
                    buf = torch.empty(int(nel), dtype=torch.float32, device=L.device).normal_()
                    bufs.append(buf)
                    args_r[ip] = same[a[ip]] = buf.data_ptr()
                for ix in nulled:
                    args_r[ix] = None
                for ix in scratch:                          # mean / rstd outputs: not into the learner's own state
                    scr = torch.empty(c, dtype=torch.float32, device=L.device)
                    bufs.append(scr)
                    args_r[ix] = scr.data_ptr()
                variants.append(tuple(args_r))
            fn = getattr(dll, n)
            for v in variants:
                fn(*v)
            reps_k = max(20, 2 * copies)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(L.stream)
            for r in range(reps_k):
                fn(*variants[r % copies])
            e1.record(L.stream)
            e1.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / reps_k
            del variants, bufs
            half = int(nbytes // 8)
            csrc = [torch.empty(half, dtype=torch.float32, device=L.device).normal_() for _ in range(copies)]
            cdst = [torch.empty(half, dtype=torch.float32, device=L.device) for _ in range(copies)]
            for a_, b_ in zip(csrc, cdst):
                b_.copy_(a_)
            cus_ = []
            for _ in range(3):
                e0.record(L.stream)
                for r in range(reps_k):
                    cdst[r % copies].copy_(csrc[r % copies])
                e1.record(L.stream)
                e1.synchronize()
                cus_.append(1e3 * e0.elapsed_time(e1) / reps_k)
            cus = sorted(cus_)[1]
            del csrc, cdst
            layers.append({"entry": n[6:], "rows,C": [rows, c], "reads_x_dy_twice": bool(two_pass), "us": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 2),
                           "frac_of_8TBps": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 3), "rotating_copies": copies,
                           "copy_same_bytes_us": round(cus, 2), "frac_of_copy": round(cus / us, 3)})
            d_ = fams.setdefault(fam, {"us_per_step": 0.0, "launches_per_step": 0, "algorithmic_MB_per_step": 0.0, "copy_same_bytes_us_per_step": 0.0})
            d_["us_per_step"] += us
            d_["launches_per_step"] += 1
            d_["algorithmic_MB_per_step"] += nbytes / 1e6
            d_["copy_same_bytes_us_per_step"] += cus
    tot = {"us_per_step": 0.0, "launches_per_step": 0, "algorithmic_MB_per_step": 0.0, "copy_same_bytes_us_per_step": 0.0}
    for d_ in fams.values():
        for k_ in tot:
            tot[k_] += d_[k_]
    fams["all_batchnorm"] = tot
    for d_ in fams.values():
        d_["frac_of_8TBps"] = d_["algorithmic_MB_per_step"] * 1e6 / (d_["us_per_step"] * 1e-6) / 1e9 / HBM_PEAK_GBS if d_["us_per_step"] else None
        d_["frac_of_copy"] = d_["copy_same_bytes_us_per_step"] / d_["us_per_step"] if d_["us_per_step"] else None
    fams["all_batchnorm"]["method"] = ("C-ABI calls of one eager inner step re-issued cold (rotating copies, > 320 MB between two touches); launches = "
                                       "API calls (a backward call without the producer's stage 1 is two kernels: reduce + apply)")
    fams["per_launch"] = layers
    return fams


def main():
    import contextlib
    argv = sys.argv[1:]
    args = parse(argv)
    if check_world(args) == "launch":
        sys.exit(launch_ranks(args, argv))
    with contextlib.redirect_stdout(sys.stderr):   # stdout carries exactly ONE JSON line
        out = _run(args)
    if out is not None:
        print(json.dumps(out))


def _run(args):
    aug_pool = None
    if args.augment and args.augment_on_host and args.augment_workers != 0:   # forked workers: before anything initialises the GPU
        from mliis_amd.augment import AugmentPool
        aug_pool = AugmentPool(None if args.augment_workers < 0 else args.augment_workers)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = "nccl"
    if torch.cuda.device_count() == 0:        # (counting devices does not initialise the GPU)
        raise SystemExit("bench.py needs an MI355X; there is no CPU path (the CPU oracle is only the reported baseline)")
    if world > 1:
        import torch.distributed as dist
        # MLIIS_DIST_BACKEND=gloo is a test hook: it lets the whole N > 1 flow (task sharding, the one all-reduce per meta-step, the
        # max-over-ranks timing, rank 0's line) be exercised by several ranks that SHARE the GPU of a one-GPU box, where RCCL refuses
        # duplicate devices.  The driver's runs use nccl (= RCCL).
        backend = os.environ.get("MLIIS_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path (the CPU oracle is only the reported baseline)")
    device = torch.device("cuda", local)
    from mliis_amd.learner import Learner
    from mliis_amd.metaseg import DeviceTask, synthetic_task
    from mliis_amd.reptile import FOMLIS, Gecko, Dist

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)
    elif rank == 0 and world > 1 and os.environ.get("MLIIS_BENCH_CPU_BASELINE"):
        cpu = json.loads(os.environ["MLIIS_BENCH_CPU_BASELINE"])
    elif rank == 0 and world > 1 and args.cpu_baseline_from:
        src = json.load(open(args.cpu_baseline_from))
        src = src.get("parsed", src) if isinstance(src, dict) else {}
        if isinstance(src.get("cpu_baseline"), dict):
            cpu = dict(src["cpu_baseline"], carried_from=os.path.basename(args.cpu_baseline_from) + " (N = 1 line; not re-timed in this run)")

    shots = 10 if args.foml else args.shots
    L = Learner(feature_extractor_name=args.backbone, image_size=args.image_size, rsd=(2, 4), learning_rate=1e-3, optimizer="adam" if args.adam else "sgd", dice=False, l2=False, seed=0, device=device,
                use_graph=not args.no_graph, max_shots=max(16, shots), spatial_pyramid_pooling=args.aspp, matmul_precision=args.precision,
                skip_decoding=args.skip_decoding, augment_batch_capacity=16 if (args.augment and not args.augment_on_host) else 0,
                rng_stream=rank)
    lanes = [Learner(feature_extractor_name=args.backbone, image_size=args.image_size, rsd=(2, 4), learning_rate=1e-3, optimizer="adam" if args.adam else "sgd", dice=False,
                     l2=False, seed=k, device=device, use_graph=not args.no_graph, max_shots=max(16, shots),
                     spatial_pyramid_pooling=args.aspp, skip_decoding=args.skip_decoding, matmul_precision=args.precision, rng_stream=rank)
             for k in range(1, args.concurrent_tasks)]
    tasks = []
    for i in range(args.pool):
        x, y = synthetic_task(shots, args.image_size, seed=1000 * rank + i)
        tasks.append(DeviceTask("synthetic_%d_%d" % (rank, i), torch.from_numpy(x).to(device), torch.from_numpy(y).to(device)))
    D = Dist()
    if args.foml:
        aug_mode = False if not args.augment else (True if args.augment_on_host else "device")
        meta = FOMLIS(L, train_shots=shots, tail_shots=5, dist=D, rng_mode="per_task", seed=0, augment=aug_mode, aug_rate=0.5, aug_pool=aug_pool,
                      lanes=lanes)
    else:
        aug_mode = False if not args.augment else (True if args.augment_on_host else "device")
        meta = Gecko(L, dist=D, rng_mode="per_task", seed=0, augment=aug_mode, aug_rate=0.5, aug_pool=aug_pool, lanes=lanes)

    def step():
        meta.train_step(tasks, num_shots=shots, inner_batch_size=args.inner_batch, inner_iters=args.inner_iters, replacement=False,
                        meta_step_size=0.1, meta_batch_size=world * args.tasks_per_gpu)

    for _ in range(args.warmup):
        step()
    D.timed = world > 1     # events around the ONE all-reduce of every timed meta-step (dist.allreduce_ms / dist.tasks_imbalance)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    D.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ar_ms = adapt_max = adapt_min = None
    if world > 1:
        import torch.distributed as dist
        D.timed = False
        inside, between = D.timing()
        tt = torch.tensor([dt, inside or 0.0, between or 0.0, -(between or 0.0)], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, ar_ms, adapt_max, adapt_min = float(tt[0].item()), float(tt[1].item()), float(tt[2].item()), -float(tt[3].item())
    imgs_per_task = ((args.inner_iters - 1) * args.inner_batch + 5) if args.foml else args.inner_iters * args.inner_batch
    value = world * args.tasks_per_gpu * imgs_per_task * args.steps / dt
    loss = L.loss_value()

    roof = dwr = fam = bnr = None
    native = None
    if rank == 0 and not args.no_roofline:
        roof, dwr, fam, bnr = roofline(L, args)
        t_step = dt / args.steps / (args.tasks_per_gpu * args.inner_iters)          # seconds per inner step (HIP-graph replay, as timed)
        issued, hbm = roof.pop("_issued_flops_per_step"), roof.pop("_hbm_bytes_per_step")
        default_cfg = (args.backbone == "efficientnet-b0" and args.image_size == 224 and args.inner_batch == 8 and not (args.aspp or args.skip_decoding))
        algo = 95.9e9 if default_cfg else None   # fwd + bwd-data + bwd-filter of the reference graph at config 2
        roof.update({
            "step_us": 1e6 * t_step,
            "step_tflops": (algo / t_step / 1e12) if algo else None,
            "step_frac_of_mfma_peak": (algo / t_step / 1e12 / MFMA_F32_PEAK_TFLOPS) if algo else None,
            "step_tflops_issued": issued / t_step / 1e12,
            "step_frac_of_mfma_peak_issued": issued / t_step / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "step_hbm_frac": (hbm / t_step / 1e9 / HBM_PEAK_GBS) if hbm else None,
            "step_note": "whole inner step: algorithmic 95.9 GFLOP of the reference graph (SURVEY.md Appendix A) and the conv flops this build "
                         "issues (pooled branch folded into a bias) / the measured step time, against the fp32 MFMA peak %.1f TFLOP/s; HBM bytes per "
                         "step from the committed rocprofv3 --pmc passes / step time / 8 TB/s" % MFMA_F32_PEAK_TFLOPS})
    if rank == 0 and args.precision == "fp32" and L.x3 is not None and not args.no_native_retime:
        native = retime_native(args, device, tasks, shots, imgs_per_task)
    if rank == 0:
        out = {
            "metric": "inner-loop images/sec (EfficientLab-6-3, 224x224, 5-shot)", "value": value, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": ("f32 (the long-K decoder convs and their filter gradients with every operand split exactly into three bf16 terms on the "
                                          "matrix cores, 6 term products, f32 accumulate: fp32-equivalent products; everything else the fp32 instruction)"
                                          if L.x3 is not None else "f32 (native fp32 matrix instruction everywhere)") if args.precision in ("fp32", "fp32-native") else ("bf16 matrix-core operands, f32 accumulate / tensors" if args.precision == "bf16" else
                                                              "bf16 matrix-core operands and bf16 expanded MBConv tensors in HBM (z0, z1, a1 and their gradients), f32 accumulate / statistics / block tensors / weights" if args.precision == "bf16-storage" else
                                                              "fp8 e4m3 operands on the 1x1 forward convs (bf16 elsewhere), f32 accumulate / tensors"), "data": "synthetic",
            "config": {"workload": "%s + %sRSD(4)+RSD(2)), %dx%d, meta-batch=%d (1 task/GPU), "
                                   "%d-shot x %d inner SGD steps of batch %d (%d image fwd+bwd+BN-EMA+SGD per task), %s outer update, fp32 tensors, "
                                   "CE loss, drop-connect on%s" % ("EfficientLab-6-3 (EfficientNet-B0 blocks 0-10" if args.backbone == "efficientnet-b0" else
                                                                 "EfficientLab with the EfficientNet-B3 encoder (blocks 0-17",
                                                                 ("ASPP + " if args.aspp else "") + ("DeepLabv3+-style skip decoder + " if args.skip_decoding else ""), args.image_size, args.image_size, world, shots,
                                                                 args.inner_iters,
                                                                 args.inner_batch, imgs_per_task, "FOMAML(tail 5)" if args.foml else "Reptile",
                                                                 ("" if args.precision in ("fp32", "fp32-native") else ", %s matrix-core operands" % args.precision) + (", Adam(beta1=0) inner optimizer" if args.adam else "") +
                                                                 ((", augmentation (aug_rate 0.5, %s)" % (("pixels on the host, %s" % ("%d worker processes" % aug_pool.workers if aug_pool else "inline")) if args.augment_on_host else "pixels on the device"))
                                                                  if args.augment else "") +
                                                                 (", %d tasks per GPU and meta-step" % args.tasks_per_gpu if args.tasks_per_gpu != 1 else "") +
                                                                 (", %d adapted concurrently" % args.concurrent_tasks if args.concurrent_tasks != 1 else "")),
                       "hip_graph": not args.no_graph, "final_loss": loss},
            "roofline": roof, "cpu_baseline": cpu,
            # what the first real multi-GPU run can be checked against: ranks, collective library, bytes of the ONE all-reduce(sum) per
            # meta-step (flat task delta + BN moving-average contributions, fp32)
            "dist": {"world": world, "backend": ("nccl (RCCL)" if backend == "nccl" else backend + " (test hook, ranks share a GPU)") if world > 1 else None, "rccl_version": _rccl_version(),
                     "allreduce_bytes_per_meta_step": int(meta._comm.numel() * 4) if meta._comm is not None else None,
                     "tasks_per_meta_step": world * args.tasks_per_gpu,
                     # decomposition of a meta-step of an N > 1 run (HIP events on the learner's stream; max over ranks): time inside the
                     # collective (it contains the wait for the slowest rank) and each rank's own work between two collectives
                     "allreduce_ms": ar_ms, "adapt_ms_slowest_rank": adapt_max, "adapt_ms_fastest_rank": adapt_min,
                     "tasks_imbalance": ((adapt_max - adapt_min) / adapt_max) if adapt_max else None},
        }
        if dwr is not None:
            out["depthwise_hbm"] = dwr
            out["kernel_families_eager_us"] = fam
        if bnr is not None:
            out["batchnorm_hbm"] = bnr
        if native is not None:
            out["fp32_native_images_per_s"] = native["value"]
            out["fp32_native_note"] = native["note"]
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
