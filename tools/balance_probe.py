"""Diagnostic: what would a PERFECTLY BALANCED launch cost?  The step launch lasts as long as its slowest wave; the launch order (lpt_sort)
spreads the heavy windows and starts them first.  Here every env of the batch replays the SAME window - all 4096 envs start on the same
second, so a launch has no tail by construction - first with the bench's random actions, then with one action row for all: the time of such
a launch, averaged over the same windows the bench's envs meet one by one, is what NO dispatch order of this kernel can beat.
usage: python tools/balance_probe.py [n_envs]      (PROF_LIB=build/variants/prof.so: the stamp build's wave cycles as well)"""
import ctypes as C, os, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K, SKIP = 330, 30
dev = torch.device("cuda", 0)
stream = bench.build_stream(100)
prof = _capi.Binding(ROOT / os.environ["PROF_LIB"]) if os.environ.get("PROF_LIB") else None
rng = np.random.default_rng(2)
acts_random = torch.from_numpy(rng.uniform(0, 10, size=(K, N, 4))).to(dev)
acts_same = acts_random[:, :1, :].expand(K, N, 4).contiguous()
random_starts = bench.workload_starts("random", N, 1, 1)[0]
for name, starts, acts in (("the bench's batch (random start seconds, random actions)", random_starts, acts_random),
                           ("every env on the same second, random actions", None, acts_random),
                           ("every env on the same second, one action row for all", None, acts_same)):
    per_start, cyc = [], []
    for s0 in ([None] if starts is not None else [36_003, 36_017, 36_031, 36_044, 36_058]):  # five start seconds: K steps from each cover 33 s of the day
        st = starts if starts is not None else np.full(N, s0 * 1_000_000, dtype=np.int64)
        cfg = bench.workload_cfg(_capi, "random", N)
        sim = _capi.Sim(prof, cfg) if prof else _lib.new_sim(cfg)
        sim.add_stream(stream); sim.set_portfolio(0, 10**12)
        sim.reset(st, 0)
        obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
        reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
        out = np.zeros((N, 8), dtype=np.uint64)
        if prof: prof.lib.lobsim_prof_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        for t in range(SKIP): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
        sim.sync()
        if prof:
            prof.lib.lobsim_prof_read(sim.h, out.ctypes.data, 1)
            for t in range(SKIP, SKIP + 100):
                sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
                prof.lib.lobsim_prof_read(sim.h, out.ctypes.data, 1)
                tot = out[:, :6].astype(np.float64).sum(axis=1)
                cyc.append((tot.mean(), tot.max()))
        else:
            t0 = time.perf_counter()
            for t in range(SKIP, K): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
            sim.sync()
            per_start.append((time.perf_counter() - t0) / (K - SKIP) * 1e6)
        assert not sim.status().any()
        sim.close()
    if prof:
        c = np.array(cyc)
        print(f"{name}: mean wave {c[:, 0].mean():.0f} cycles, slowest wave of a launch {c[:, 1].mean():.0f} (stamp build, {len(c)} launches)", flush=True)
    else:
        print(f"{name}: {np.mean(per_start):.2f} us per {N}-env step ({', '.join('%.2f' % x for x in per_start)}) -> {N / np.mean(per_start):.1f} M env-steps/s", flush=True)
