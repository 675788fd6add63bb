#!/usr/bin/env python3
"""Stand-alone reproducer attempt (PyTorch only, nothing of this repository inside the graph): do reductions captured in a hipGraph
(torch.isfinite(t).all() -- elementwise temporaries from the graph's memory pool + a two-stage reduction with its semaphore memset) give wrong
answers at replay when eager work runs between replays?  That is the pattern under which the graphed training step's tracer (HAVATAR_NAN_TRACE=1)
raises flags on tensors that are demonstrably finite (tools/graph_anomaly_hunt.py, DESIGN.md "graphed training step").

N_OPS chained element-wise ops on fresh tensors (so the pool recycles blocks like a training step does), each followed by isfinite().all() and,
every fourth op, a sum() whose value is known.  Between replays (every fourth): EAGER = none | mul (element-wise kernels only) | alloc (allocator
traffic only) | sum (ONE torch.sum of a persistent tensor) | smallsum (a one-block sum) | elementwise (mul + sum) | conv (MIOpen convolutions +
max).  Prints how many flags / sums were wrong per replay.  CAPTURE_STREAM=side|own as in harness/train.py; REPLAY_ON=default|side.

Measured on MI355X, ROCm 7.0.2 runtime / PyTorch 2.10.0+rocm7.0 (profiles/r05_graph_anomaly_root_cause.txt): none / mul / alloc: 0 of 60 replays
wrong; sum / smallsum / elementwise / conv: 56 of 60 (every replay after the first eager reduction; the same graph nodes in every process: the
sum of ones of op 76 returns 0x01010101 -- the bytes a bool temporary left in its output block, i.e. the node did not write its result -- and
the is-finite flag of op 77 reads false), whatever stream captures or replays.  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (WORKAROUND=inprocess sets it
the way havatar_amd/__init__.py does): 0 of 60 in every run."""
import os
import sys

import torch

if os.environ.get("WORKAROUND") == "inprocess":          # what havatar_amd/__init__.py does: set after `import torch`, before the first HIP call
    assert not torch.cuda.is_initialized()
    os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "0"
dev = torch.device("cuda:0")
N_OPS, REPLAYS, EAGER = int(os.environ.get("N_OPS", "150")), int(os.environ.get("REPLAYS", "60")), os.environ.get("EAGER", "conv")
shapes = [(2, 512, 65, 65), (2, 256, 129, 129), (2, 512, 32, 32), (512, 512, 3, 3), (2, 256, 128, 128), (512, 64, 68)]
torch.manual_seed(0)
x0 = [torch.rand(s, device=dev) + 0.5 for s in shapes]
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
img = torch.randn(1, 64, 128, 128, device=dev)
eager_t = torch.randn(1 << 22, device=dev)
REPLAY_ON = os.environ.get("REPLAY_ON", "default")          # default | side: the stream the replays are launched on


def body():
    flags, sums, want = [], [], []
    cur = [t for t in x0]
    for i in range(N_OPS):
        k = i % len(shapes)
        a = cur[k] * 1.0 + 0.0          # a fresh tensor in the pool, all ones-ish, finite
        flags.append(torch.isfinite(a).all())
        if i % 4 == 0:
            ones = torch.ones_like(a)
            sums.append(ones.sum())
            want.append(float(a.numel()))
        cur[k] = a
    return flags, sums, want


def eager_work():
    if EAGER == "none":
        return
    if EAGER == "elementwise":
        t = torch.randn(1 << 22, device=dev)
        for _ in range(20):
            t = t * 1.01 + 0.1
        return float(t.sum())
    if EAGER == "mul":                    # element-wise kernels only, no reduction; the sync is explicit
        t = torch.randn(1 << 22, device=dev)
        for _ in range(20):
            t = t * 1.01 + 0.1
        torch.cuda.synchronize()
        return
    if EAGER == "sum":                    # one two-stage reduction (buffer + semaphore memset) on a persistent tensor, nothing else
        return float(eager_t.sum())
    if EAGER == "smallsum":               # a single-block reduction: no semaphore, no memset
        return float(eager_t[:256].sum())
    if EAGER == "alloc":                  # allocator traffic only, no kernel
        junk = [torch.empty(1 << n, device=dev) for n in (8, 12, 16, 20, 24)]
        del junk
        torch.cuda.synchronize()
        return
    with torch.no_grad():
        y = img
        for _ in range(6):
            y = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(y, w, padding=1), 0.2)
        return float(y.abs().max())


side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    body()                                # warm-up on the capture stream
torch.cuda.current_stream(dev).wait_stream(side)
eager_work()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side if os.environ.get("CAPTURE_STREAM", "side") == "side" else None, capture_error_mode="thread_local"):
    flags, sums, want = body()
bad_replays, bad_total = 0, 0
for r in range(REPLAYS):
    if REPLAY_ON == "side":
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            g.replay()
    else:
        g.replay()
    torch.cuda.synchronize()
    wrong_f = [i for i, f in enumerate(flags) if not bool(f)]
    wrong_s = [(4 * i, float(s), v) for i, (s, v) in enumerate(zip(sums, want)) if float(s) != v]
    nf, ns = len(wrong_f), len(wrong_s)
    if (nf or ns) and bad_replays < 3:
        print("   wrong flags at ops %s; wrong sums (op, got, want): %s" % (wrong_f[:8], wrong_s[:4]), flush=True)
    if nf or ns:
        bad_replays += 1
        bad_total += nf + ns
        if bad_replays <= 10:
            print("replay %d (%s eager work before it): %d of %d is-finite flags false, %d of %d sums wrong" % (
                r, "with" if (r % 4 == 0 and r) else "no", nf, len(flags), ns, len(sums)), flush=True)
    if r % 4 == 3:
        eager_work()
print("repro_graph_reduce: EAGER=%s N_OPS=%d: %d of %d replays had a wrong flag or sum (%d wrong values)" % (EAGER, N_OPS, bad_replays, REPLAYS, bad_total))
