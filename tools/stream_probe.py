"""Diagnostic: what a plain streaming kernel reaches on this GPU (read-only reduction and copy over a few GB, torch's own kernels) - the practical
ceiling next to the 8 TB/s peak that bench.py's roofline fractions are quoted against.  usage: python tools/stream_probe.py"""
import torch
dev = torch.device("cuda", 0)
for gib in (1, 4, 8):
    n = gib * 2**30 // 8
    a = torch.ones(n, dtype=torch.float64, device=dev)
    b = torch.empty_like(a)
    for name, fn, nbytes in (("sum (read)", lambda: a.sum(), 8 * n), ("copy (read + write)", lambda: b.copy_(a), 16 * n)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{gib} GiB f64 {name}: {ms:.3f} ms -> {nbytes / ms / 1e9:.2f} TB/s", flush=True)
    del a, b
