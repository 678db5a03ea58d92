"""Diagnostic: how well can a weight computed from the STREAM ALONE predict what a window costs its wave?  (-DLOBSIM_PROFILE stamp build.)
Per env and launch: the wave's cycles (stamps) against features of the window it replayed, counted on the host from the packed stream -
messages by type, runs of executions, removals, limit orders at prices the window has not seen - least squares, and what a launch would last
under list scheduling (4 waves a SIMD, heaviest first) with each candidate weight against the true cycles (the bound no predictor can beat).
usage: PROF_LIB=build/variants/prof.so python tools/weight_fit.py [n_envs]   -> prints the fits; LOBSIM_NO_LPT=1 keeps the launches in env order"""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi

b = _capi.Binding(ROOT / os.environ["PROF_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K, SKIP = 140, 20
cfg = bench.workload_cfg(_capi, "random", N)
sim = _capi.Sim(b, cfg)
stream = bench.build_stream(60)
sim.add_stream(stream); sim.set_portfolio(0, 10**12)
starts, sids = bench.workload_starts("random", N, 1, 1)
sim.reset(starts, sids)
dev = torch.device("cuda", 0)
acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(K, N, 4))).to(dev)
obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
out = np.zeros((N, 8), dtype=np.uint64)
b.lib.lobsim_prof_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
t_us, meta = stream.msgs[:, 0], stream.msgs[:, 3]
typ = (meta >> 48) & 0xf
is_mkt, is_lim, is_rem = typ == 4, typ == 1, (typ == 2) | (typ == 3)
run_start = is_mkt & ~np.concatenate([[False], is_mkt[:-1]])
cum = lambda x: np.concatenate([[0], np.cumsum(x)])
c_all, c_mkt, c_run, c_lim, c_rem = cum(np.ones(len(t_us))), cum(is_mkt), cum(run_start), cum(is_lim), cum(is_rem)
# a removal right behind an execution (the order the sweep consumed is gone: a miss of the position cache, a queue scan)
rem_after_mkt = is_rem & np.concatenate([[False], is_mkt[:-1]])
c_ram = cum(rem_after_mkt)
X, Y, PREV = [], [], []
prev_tot = None
for t in range(K):
    sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    b.lib.lobsim_prof_read(sim.h, out.ctypes.data, 1)
    tot = out[:, :6].astype(np.float64).sum(axis=1)
    if t >= SKIP:
        lo = np.searchsorted(t_us, starts + t * 100_000, side="right"); hi = np.searchsorted(t_us, starts + (t + 1) * 100_000, side="right")
        sec = ((starts + (t + 1) * 100_000) % 1_000_000 == 0).astype(np.float64)
        X.append(np.column_stack([c_all[hi] - c_all[lo], c_run[hi] - c_run[lo], c_mkt[hi] - c_mkt[lo], c_rem[hi] - c_rem[lo], c_lim[hi] - c_lim[lo], c_ram[hi] - c_ram[lo], sec]))
        Y.append(tot); PREV.append(prev_tot)
    prev_tot = tot
X, Y, PREV = np.array(X), np.array(Y), np.array(PREV)  # [T, N, F], [T, N]
# (alignment check: the correlation of a window's message count with the wave's cycles must peak at shift 0)
print("corr(msgs of window t + shift, cycles of launch t): " + ", ".join(f"{sh:+d}: {np.corrcoef(X[max(0, sh):len(X) + min(0, sh), :, 0].reshape(-1), Y[max(0, -sh):len(Y) - max(0, sh)].reshape(-1))[0, 1]:.3f}" for sh in (-2, -1, 0, 1, 2)))
names = ["msgs", "exec runs", "exec msgs", "removals", "limits", "removal behind an execution", "second boundary"]
flat_x, flat_y = X.reshape(-1, X.shape[2]), Y.reshape(-1)
A = np.column_stack([flat_x, np.ones(len(flat_y))])
coef = np.linalg.lstsq(A, flat_y, rcond=None)[0]
pred = A @ coef
print(f"{N} envs x {len(Y)} launches; wave cycles mean {flat_y.mean():.0f} std {flat_y.std():.0f}")
print("least squares over the stream's features: " + ", ".join(f"{c:.0f} x {n}" for c, n in zip(coef, names + ["1"])) + f";  R^2 {1 - ((flat_y - pred) ** 2).sum() / ((flat_y - flat_y.mean()) ** 2).sum():.3f}")
w_now = flat_x[:, 0] + 8 * flat_x[:, 1]
print(f"correlation with the wave's cycles: msgs + 8 runs (the table's weight) {np.corrcoef(w_now, flat_y)[0, 1]:.3f}; msgs {np.corrcoef(flat_x[:, 0], flat_y)[0, 1]:.3f}; the fit {np.corrcoef(pred, flat_y)[0, 1]:.3f}; "
      f"the env's own cycles one launch earlier {np.corrcoef(PREV.reshape(-1), flat_y)[0, 1]:.3f}")


def makespan(weights, true, per=4):
    """list scheduling of one launch: sort by weight descending, deal round-robin to N / per SIMDs; a SIMD's waves run concurrently, each
    slowed by its mates - modelled crudely as: the SIMD ends when its longest wave does, lengthened by a share of the others' work."""
    order = np.argsort(-weights, kind="stable")
    n_simd = len(true) // per
    simd = np.empty(len(true), dtype=np.int64); simd[order] = np.arange(len(true)) % n_simd
    longest = np.zeros(n_simd); total = np.zeros(n_simd)
    np.maximum.at(longest, simd, true); np.add.at(total, simd, true)
    return (longest + 0.12 * (total - longest)).max()


for nm, wfun in (("env order", None), ("msgs + 8 runs", lambda x, p: x[:, 0] + 8 * x[:, 1]), ("the fit", lambda x, p: np.column_stack([x, np.ones(len(x))]) @ coef),
                 ("the fit + 0.3 x the env's previous cycles", lambda x, p: np.column_stack([x, np.ones(len(x))]) @ coef + 0.3 * p), ("the TRUE cycles (the bound)", "true")):
    ms = []
    for k in range(len(Y)):
        if wfun is None: w = -np.arange(N, dtype=np.float64)
        elif wfun == "true": w = Y[k]
        else: w = wfun(X[k], PREV[k])
        ms.append(makespan(w, Y[k]))
    print(f"  modelled launch under '{nm}': {np.mean(ms):.0f} cycles (slowest wave alone: {Y.max(axis=1).mean():.0f})")
