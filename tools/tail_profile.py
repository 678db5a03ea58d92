"""Diagnostic: what makes the slowest wave of each launch slow?  (-DLOBSIM_PROFILE build, s_memtime stamps.)
A launch lasts as long as its busiest env, so the step rate is set by the per-launch MAX over envs, not the mean."""
import ctypes as C, subprocess, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi, _lib

import os
so = ROOT / "gpurun_out" / "liblobsim_prof.so"
so.parent.mkdir(exist_ok=True)
if os.environ.get("PROF_LIB"): so = ROOT / os.environ["PROF_LIB"]  # prebuilt in the build container: tools/build_variant.sh prof -DLOBSIM_PROFILE
else: _lib.compile_library(so, ["-DLOBSIM_PROFILE", *(["-DLOBSIM_PROFILE_MSG"] if "msg" in sys.argv else [])])
b = _capi.Binding(so)
WL = os.environ.get("TP_WORKLOAD", "random")  # a bench.py workload: random / ad / rs / deep50 (TP_GENERIC=1: the generic kernels)
if os.environ.get("TP_GENERIC"): os.environ["LOBSIM_GENERIC_KERNELS"] = "1"
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else (8192 if WL == "deep50" else 4096)
K = int(os.environ.get("TP_K", 120))
SKIP = int(os.environ.get("TP_SKIP", 20))  # launches before the ones that are analysed (TP_SKIP=500 TP_K=600: steps 500 - 600 after the reset)
cfg = bench.workload_cfg(_capi, WL, N, agent_orders=int(os.environ.get("TP_AGENT_ORDERS", 0)))
sim = _capi.Sim(b, cfg)
streams = bench.build_streams(WL, max(60, K // 10 + 10))
for st in streams: sim.add_stream(st)
sim.set_portfolio(0, 10**12)
starts, sids = bench.workload_starts(WL, N, len(streams), 1)
sim.reset(starts, sids)
dev = torch.device("cuda", 0)
acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(min(K, 200), N, 4))).to(dev)
obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
out = np.zeros((N, 8), dtype=np.uint64)
b.lib.lobsim_prof_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
b.lib.lobsim_prof_read_messages.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
tus_of = [st.msgs[:, 0] for st in streams]
names = ["load_env", "build_ops", "agent_ops", "messages+resync", "features+reward", "outputs+store"]
rows = []
for t in range(K):
    sim.step_device(acts[t % acts.shape[0]].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    b.lib.lobsim_prof_read(sim.h, out.ctypes.data, 1)
    evb = np.zeros((N, 16), dtype=np.uint64)
    if "msg" not in sys.argv: b.lib.lobsim_prof_read_messages(sim.h, evb.ctypes.data, 1)
    if t < SKIP: continue
    ph = out[:, :6].astype(np.float64)
    tot = ph.sum(axis=1)
    lo, hi = starts + t * 100_000, starts + (t + 1) * 100_000
    nmsg = np.zeros(N, dtype=np.int64)
    for k, tus in enumerate(tus_of):
        mk = sids == k
        nmsg[mk] = np.searchsorted(tus, hi[mk], side="right") - np.searchsorted(tus, lo[mk], side="right")
    sec_cross = (hi // 1_000_000) != (lo // 1_000_000)
    order = np.argsort(tot)
    i = order[-1]
    rows.append(dict(max=tot[i], mean=tot.mean(), p50=np.median(tot), p99=np.percentile(tot, 99), p999=np.percentile(tot, 99.9), ph_max=ph[i], ph_mean=ph.mean(axis=0), n_max=nmsg[i], n_mean=nmsg.mean(), n_top=nmsg.max(),
                     cpm=ph[:, 3] / np.maximum(nmsg, 1), nmsg=nmsg.copy(), msgc=ph[:, 3].copy(), tot=tot.copy(), sec=sec_cross.copy(), ops=out[:, 6].astype(float), ev=evb[:, :6].astype(float), ev_max=evb[i, :6].astype(float), sub=evb[:, 6:16].astype(float).mean(axis=0), sub_max=evb[i, 6:16].astype(float)))
if "rows" in sys.argv:  # the slowest env of each launch, one line per launch
    print("slowest env per launch: total | load build agent msgs feat out | n_msgs | chunks compactions ladder_ops swept_levels rare serial_agent_ops | ser_rem unknown cyc_parallel cyc_serial_msgs ser_lim ser_mkt segments cyc_runs cyc_to_first_pass cyc_after_window")
    for r in rows[:40]:
        print("%6d | %s | %2d | %s | %s" % (r["max"], " ".join("%5d" % x for x in r["ph_max"]), r["n_max"], " ".join("%2d" % x for x in r["ev_max"]), " ".join("%2d" % x if k not in (2, 3, 7, 8, 9) else "%5d" % x for k, x in enumerate(r["sub_max"][:10]))))
A = lambda k: np.array([r[k] for r in rows])
print(f"workload {WL}{' (generic kernels)' if os.environ.get('TP_GENERIC') else ''}, limits {sim.limits()}")
print(f"{N} envs, {len(rows)} launches (steps {SKIP} .. {K} after the reset): wave cycles  mean {A('mean').mean():.0f}  p50 {A('p50').mean():.0f}  p99 {A('p99').mean():.0f}  p99.9 {A('p999').mean():.0f}  MAX {A('max').mean():.0f}")
print(f"messages in the slowest env {A('n_max').mean():.1f} (mean env {A('n_mean').mean():.1f}, busiest env {A('n_top').mean():.1f})")
pm, pa = np.array([r['ph_max'] for r in rows]).mean(axis=0), np.array([r['ph_mean'] for r in rows]).mean(axis=0)
for k, n in enumerate(names): print(f"  {n:18s} slowest env {pm[k]:8.0f}   mean env {pa[k]:8.0f}")
nm, mc, tt, sc, ops = np.concatenate(A('nmsg')), np.concatenate(A('msgc')), np.concatenate(A('tot')), np.concatenate(A('sec')), np.concatenate(A('ops'))
fit = np.polyfit(nm, mc, 1)
print(f"message phase cycles ~ {fit[0]:.0f} * n_msgs + {fit[1]:.0f};  residual std {np.std(mc - np.polyval(fit, nm)):.0f}")
res = mc - np.polyval(fit, nm)
print(f"message-phase residual percentiles 50/99/99.9/max: {np.percentile(res, [50, 99, 99.9, 100]).round(0).tolist()}")
print(f"second-boundary steps: share {sc.mean():.3f}; mean total cycles with {tt[sc].mean():.0f} / without {tt[~sc].mean():.0f}")
other = tt - mc
fo = np.polyfit(ops, other, 1)
print(f"non-message cycles ~ {fo[0]:.0f} * agent_ops + {fo[1]:.0f}; percentiles 50/99/max {np.percentile(other, [50, 99, 100]).round(0).tolist()}; agent ops mean {ops.mean():.1f} max {ops.max():.0f}")
ev = np.concatenate(A('ev')); evm = np.array([r['ev_max'] for r in rows])
evn = ["slab chunks read", "compactions", "ladder inserts+removes", "levels swept by executions", "limits opening a new best", "agent ops left to the serial path"]
print("events per env-step (whole step: agent ops + messages):")
for k, n in enumerate(evn): print(f"  {n:28s} mean env {ev[:, k].mean():7.2f}   slowest env {evm[:, k].mean():7.2f}")
X = np.column_stack([nm, ev[:, 0], ev[:, 1], ev[:, 2], np.ones(len(nm))])
coef = np.linalg.lstsq(X, mc, rcond=None)[0]
print("message-phase cycles ~ " + " + ".join(f"{c:.0f}*{n}" for c, n in zip(coef, ["msgs", "chunks", "compactions", "ladder_ops", "1"])))
sub = np.array([r["sub"] for r in rows]).mean(axis=0)
subm = np.array([r["sub_max"] for r in rows]).mean(axis=0)
print("slowest env of a launch: serial removals %.2f, unknown ids %.2f (cycles in lane-parallel passes %.0f, in single serial messages %.0f), serial limits %.2f, serial markets %.2f, segments %.2f, cycles in execution runs %.0f, from the top of the window to its first pass %.0f, after the window %.0f" % tuple(subm[:10]))
print("per env-step (mean env): serial removals %.2f, of them unknown ids %.2f (cycles in lane-parallel passes %.0f, in single serial messages %.0f); serial limit messages %.2f, serial market messages %.2f; segments %.2f; cycles in execution runs %.0f, from the top of the window to its first pass %.0f, after the window %.0f" % tuple(sub[:10]))
print("op-list generation sub-phases (mean env cycles): pdf %.0f | normalise+round %.0f | inventory/best prices %.0f | agent entries -> levels %.0f | diff publish %.0f... candidates %.0f | op list %.0f" % tuple(sub[:7][[0, 1, 2, 3, 4, 4, 5]]) if False else "op-list sub-phases (mean env cycles): " + ", ".join(f"{x:.0f}" for x in sub[:6]) + f"; feature state load {sub[7]:.0f}, make_view {sub[8]:.0f}")
if "msg" not in sys.argv: sys.exit(0)
pmsg = np.zeros((N, 16), dtype=np.uint64)
b.lib.lobsim_prof_read_messages.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
b.lib.lobsim_prof_read_messages(sim.h, pmsg.ctypes.data, 0)
cls = ["limit, level exists", "limit, opens level", "limit, crossing", "cancel", "delete", "delete/cancel emptying level", "market", "other"]
cyc, cnt = pmsg[:, :8].sum(axis=0).astype(float), pmsg[:, 8:].sum(axis=0).astype(float)
print("per message class (stamps drain the memory queue before and after: upper bounds):")
for k, n in enumerate(cls):
    if cnt[k]: print(f"  {n:30s} share {100 * cnt[k] / cnt.sum():5.1f} %   {cyc[k] / cnt[k]:7.0f} cycles each   {100 * cyc[k] / cyc.sum():5.1f} % of message time")
