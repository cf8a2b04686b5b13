#!/usr/bin/env python3
"""Which pairs of torch streams run kernels concurrently on this box?  (Two HIP streams that share one HSA queue serialize.)
A one-thread spin kernel (torch.cuda._sleep) on each stream of a pair: wall time ~1x the spin if they overlap, ~2x if not."""
import json, time, torch

def pair_time(a, b, cycles=4_000_000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st in (a, b):
        with torch.cuda.stream(st):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

dev = torch.device("cuda:0")
torch.cuda._sleep(1000)
normal = [torch.cuda.Stream(device=dev) for _ in range(6)]
high = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(3)]
cur = torch.cuda.current_stream()
one = pair_time(normal[0], normal[0])
out = {"two_spins_on_one_stream_ms": one, "pairs_ms": {}}
names = {**{f"n{i}": s for i, s in enumerate(normal)}, **{f"h{i}": s for i, s in enumerate(high)}, "default": cur}
keys = list(names)
for i, a in enumerate(keys):
    for b in keys[i + 1:]:
        out["pairs_ms"][f"{a}+{b}"] = round(pair_time(names[a], names[b]), 3)
print(json.dumps(out, indent=1))
