"""would a PARTITION of the chip's CUs let the deferred filter gradients run beside the encoder's backward chain?

The backward pass of one headline inner step is cut at two points (passes.py `_segment_hook`) and the two pieces are captured as HIP
graphs of their own: the ENCODER BACKWARD chain (blocks 10 .. 0 and the stem: ~75 launches, most of them far below 256 workgroups) and
the batched FILTER GRADIENTS (8 full-chip launches).  Each graph is replayed on streams created with hipExtStreamCreateWithCUMask
(a linear graph's kernels run on the launching stream, so the mask applies -- the probe checks that: a masked replay must be slower)
alone and side by side on complementary masks.  Prints microseconds per replay; profiles/r05_cumask.txt keeps the output.

    python tools/cumask_probe.py [reps]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mliis_amd._lib import lib  # noqa: E402
from mliis_amd.learner import Learner  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(cus):
    """A stream restricted to the CUs in `cus` (indices 0..255)."""
    words = (C.c_uint32 * 8)()
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(st.value), st.value


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    precision = os.environ.get("PROBE_PRECISION", "fp32-native")   # (the split-product kernels are single-stream only: see r05_notes.md)
    dev = torch.device("cuda:0")
    L = Learner(image_size=224, seed=1, use_graph=False, drop_connect=True, matmul_precision=precision)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 255, (5, 224, 224, 3)).astype(np.float32)
    lab = (rng.uniform(size=(5, 224, 224, 1)) > 0.6).astype(np.float32)
    L.load_task(x, np.concatenate([1 - lab, lab], -1))
    idx = [0, 1, 2, 3, 4, 0, 1, 2]
    for _ in range(3):
        L.inner_step(idx)
    L.synchronize()
    graphs = {}
    state = {"open": None}

    def hook(name):
        st = L.stream.cuda_stream
        if state["open"] is not None:
            g = C.c_void_p()
            lib.call("mliis_graph_end_capture", st, C.byref(g))
            graphs[state["open"]] = g
            state["open"] = None
        if name in ("encoder_backward", "filter_gradients"):
            lib.call("mliis_graph_begin_capture", st)
            state["open"] = name

    L._segment_hook = hook
    L._capturing = True
    try:
        L.inner_step(idx)     # (the two segments are captured, not executed: this step's numbers are garbage, the buffers stay valid)
    finally:
        L._capturing = False
        L._segment_hook = None
    L.synchronize()
    chain, filt = graphs["encoder_backward"], graphs["filter_gradients"]

    def time_one(g, stream_t, stream_p):
        with torch.cuda.stream(stream_t):
            for _ in range(3):
                lib.call("mliis_graph_launch", g, stream_p)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream_t)
            for _ in range(reps):
                lib.call("mliis_graph_launch", g, stream_p)
            e.record(stream_t)
            e.synchronize()
        return s.elapsed_time(e) * 1e3 / reps

    def time_both(sa, pa, sb, pb):
        """chain on stream a, filter gradients on stream b, started together; per pair of replays."""
        torch.cuda.synchronize()
        tot = 0.0
        for _ in range(reps):
            s = torch.cuda.Event(enable_timing=True)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(sa)
            sb.wait_event(s)
            lib.call("mliis_graph_launch", chain, pa)
            lib.call("mliis_graph_launch", filt, pb)
            ea.record(sa)
            eb.record(sb)
            ea.synchronize()
            eb.synchronize()
            tot += max(s.elapsed_time(ea), s.elapsed_time(eb))
        return tot * 1e3 / reps

    full, pfull = masked_stream(range(256))
    print("precision %s; encoder-backward chain and batched filter gradients of one inner step (N = 8, 224x224), us per replay" % precision)
    base_c, base_f = time_one(chain, full, pfull), time_one(filt, full, pfull)
    print("all 256 CUs:            chain %7.1f   filter gradients %7.1f   one after the other %7.1f" % (base_c, base_f, base_c + base_f))
    # (CU ids: 32 consecutive ids per XCD.  Masks that take every k-th id instead were measured too: they run at the unmasked speed --
    #  the runtime does not honour them)
    for k in (32, 64, 80, 96, 128):
        a, b = list(range(256 - k)), list(range(256 - k, 256))
        sa, pa = masked_stream(a)
        sb, pb = masked_stream(b)
        tc, tf = time_one(chain, sa, pa), time_one(filt, sb, pb)
        tb = time_both(sa, pa, sb, pb)
        print("chain on %3d CUs %7.1f   filter gradients on %3d CUs %7.1f   side by side %7.1f   (one after the other on the whole chip %7.1f)" % (
            len(a), tc, len(b), tf, tb, base_c + base_f))
    both_full = time_both(full, pfull, masked_stream(range(256))[0], masked_stream(range(256))[1])
    print("two unmasked streams side by side: %7.1f" % both_full)
    L.close()


if __name__ == "__main__":
    main()
