"""(GPU box) the creeping one-share policy of tests/test_gpu_agent_capacity.py, stopping at the first book difference."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import bench, oracle_lib as ol
from rl4mm_amd import _capi, _lib
from test_gpu_agent_capacity import creeping_actions, _pair, T

n = 8
cfg = _capi.make_cfg(n, 10, bench.default_features(_capi), step_us=100_000, episode_us=600_000_000)
rng = np.random.default_rng(3)
starts = (36_000 + rng.integers(0, 30, n)) * 1_000_000
(gpu, _), (cpu, _) = _pair(cfg, bench.build_stream(60), starts)
acts = creeping_actions(T, n, rng)
for t in range(T):
    gpu.step(acts[t]); cpu.step(acts[t])
    for e in range(n):
        qg, qc = gpu.queues(e), cpu.queues(e)
        if qg.shape != qc.shape or not np.array_equal(qg, qc):
            print("step", t, "env", e, "shapes", qg.shape, qc.shape, "agent orders", gpu.env_state(e)["n_agent_orders"], cpu.env_state(e)["n_agent_orders"])
            if qg.shape == qc.shape:
                bad = np.flatnonzero((qg != qc).any(axis=1))
                print("rows", bad[:20])
                for b in bad[:12]:
                    print("  gpu", qg[b], "cpu", qc[b])
                p = qg[bad[0], 1]; s = qg[bad[0], 0]
                print("gpu level", gpu.level_orders(e, int(s), int(p))); print("cpu level", cpu.level_orders(e, int(s), int(p)))
            sys.exit(1)
print("no difference in", T, "steps")
