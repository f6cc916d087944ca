"""Negative control for the round-4 segfault record: what the pre-fix Branch did when the pooled
stream it had cached WAS the capture stream -- record an event on the capturing stream and make the same stream wait for it, at
the fork and again at the join -- and then replay the captured graph.  Prints CAPTURED / REPLAYED; a fault shows as the exit code."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import tripled_amd  # noqa: F401  (sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before HIP initialises)

dev = torch.device("cuda", 0)
n_self_waits = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cap = torch.cuda.Stream()
x = torch.ones(1 << 20, device=dev)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=cap):
    if n_self_waits > 0:
        cap.wait_stream(cap)          # old Branch.__enter__ with stream == current
    y = x * 2.0
    z = x + 1.0
    if n_self_waits > 1:
        cap.wait_stream(cap)          # old Branch.join
    out = y + z
print("CAPTURED", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("REPLAYED", float(out.sum()) / (1 << 20), flush=True)
