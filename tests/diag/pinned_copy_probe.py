"""Diagnostic: what reading the handle's pinned step buffers costs the host (numpy copy variants), and the step call alone."""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench  # noqa: E402
from rl4mm_amd import _capi, _lib  # noqa: E402

N = 4096
stream = bench.build_streams("random", 30)[0]
sim = _lib.new_sim(bench.workload_cfg(_capi, "random", N))
sim.add_stream(stream); sim.set_portfolio(0, 10**12)
starts, sids = bench.workload_starts("random", N, 1, 1)
sim.reset(starts, sids)
io = sim.host_io()
acts = np.random.default_rng(0).uniform(0, 10, size=(64, N, 4))


def timeit(label, fn, n=200):
    for _ in range(10):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    print(f"{label:50s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us")


k = [0]
def step():
    k[0] = (k[0] + 1) % 64
    np.copyto(io["actions"], acts[k[0]])
    sim.step_host_io(False, 1)
timeit("copyto(actions) + step_host_io (f64 obs)", step)
def step32():
    k[0] = (k[0] + 1) % 64
    np.copyto(io["actions"], acts[k[0]])
    sim.step_host_io(False, 2)
timeit("copyto(actions) + step_host_io (f32 obs)", step32)
timeit("np.copyto(pinned actions, acts)", lambda: np.copyto(io["actions"], acts[3]))
timeit("obs.copy() from pinned (320 KB)", lambda: io["obs"].copy())
pre = np.empty_like(io["obs"])
timeit("np.copyto(prealloc, pinned obs)", lambda: np.copyto(pre, io["obs"]))
timeit("ctypes.memmove(prealloc, pinned obs)", lambda: C.memmove(pre.ctypes.data, io["obs"].ctypes.data, pre.nbytes))
src = np.random.default_rng(1).uniform(size=io["obs"].shape)
timeit("src.copy() from ordinary memory (320 KB)", lambda: src.copy())
timeit("np.copyto(prealloc, ordinary)", lambda: np.copyto(pre, src))
timeit("obs_f32.copy() from pinned (160 KB)", lambda: io["obs_f32"].copy())
timeit("reward.copy() + done.copy()", lambda: (io["reward"].copy(), io["done"].copy()))
timeit("np.empty(320 KB) alone", lambda: np.empty_like(src))
sim.close()

# cold reads: right after the GPU wrote the buffers
sim2 = _lib.new_sim(bench.workload_cfg(_capi, "random", N))
sim2.add_stream(stream); sim2.set_portfolio(0, 10**12); sim2.reset(starts, sids)
io = sim2.host_io()
for want, key in ((1, "obs"), (2, "obs_f32")):
    tc = ts = 0.0
    pre = np.empty_like(io[key])
    for i in range(200):
        np.copyto(io["actions"], acts[i % 64])
        t0 = time.perf_counter()
        sim2.step_host_io(False, want)
        t1 = time.perf_counter()
        np.copyto(pre, io[key]); r = io["reward"].copy(); d = io["done"].copy()
        t2 = time.perf_counter()
        if i >= 10:
            ts += t1 - t0; tc += t2 - t1
    print(f"{key}: step_host_io {ts / 190 * 1e6:.1f} us, cold copy of outputs into a preallocated array {tc / 190 * 1e6:.1f} us")
tc = 0.0
for i in range(200):
    t0 = time.perf_counter()
    o, r, d = sim2.step(acts[i % 64])
    if i >= 10:
        tc += time.perf_counter() - t0
print(f"lobsim_step (host pointers, C memcpy both ways) {tc / 190 * 1e6:.1f} us")
import os
print("cpus allowed", len(os.sched_getaffinity(0)), "numa nodes", len([d for d in os.listdir('/sys/devices/system/node') if d.startswith('node')]))
sim2.close()
