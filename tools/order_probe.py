"""Diagnostic: what could a launch order chosen with PERFECT knowledge of every env's next window buy the per-step launch?
(An upper bound for any weight-balanced dispatch built on the device: the order costs nothing here - it is worked out on the host from the
stream and uploaded between launches, outside the measured interval.)
Build: tools/build_variant.sh order -DLOBSIM_PROFILE -DLOBSIM_PROFILE_SKEW -DLOBSIM_DIAG_ORDER -DLOBSIM_STEP_WAVES=4   (device clock at the top
and bottom of every wave, HW_ID: where it ran) and tools/build_variant.sh orderfast -DLOBSIM_DIAG_ORDER (no stamps: wall-clock per step).
usage: PROF_LIB=build/variants/order.so FAST_LIB=build/variants/orderfast.so python tools/order_probe.py [n_envs]
Policies, interleaved launch by launch:
  none      order = null (the product: workgroup b runs env b, no load of order[])
  identity  order[b] = b (what the dependent load of order[blockIdx] costs by itself)
  balanced  the envs sorted by the weight of their next window, dealt to the SIMDs in snake order (every SIMD one env of each quartile, the
            heaviest with the lightest), through the block -> SIMD placement measured on the launches before
  packed    the opposite: the four heaviest envs on one SIMD, the next four on the next ... (how much does the pairing matter at all?)
  lpt       heaviest envs in the lowest block numbers (dispatched first), no regard for the SIMD
"""
import ctypes as C, os, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 100
dev = torch.device("cuda", 0)
stream = bench.build_stream(60)
tus = stream.msgs[:, 0]
is_mkt = ((stream.msgs[:, 3] >> 48) & 0xf) == 4
run_start = is_mkt & ~np.concatenate([[False], is_mkt[:-1]])
cum_runs = np.concatenate([[0], np.cumsum(run_start)])
starts = (36_000 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(K, N, 4))).to(dev)
obs = torch.zeros((N, 16), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)


is_rem = np.isin((stream.msgs[:, 3] >> 48) & 0xf, (2, 3))
cum_mkt, cum_rem = np.concatenate([[0], np.cumsum(is_mkt)]), np.concatenate([[0], np.cumsum(is_rem)])
RUN_W = float(os.environ.get("OP_RUN_W", 8))


def weights(t, kind="runs"):  # the kernel's window_weight of step t: messages + 8 per run of executions
    lo = np.searchsorted(tus, starts + t * 100_000, side="right")
    hi = np.searchsorted(tus, starts + (t + 1) * 100_000, side="right")
    if kind == "count": return (hi - lo).astype(np.float64)
    if kind == "runs16": return (hi - lo) + 16.0 * (cum_runs[hi] - cum_runs[lo])
    if kind == "runs4": return (hi - lo) + 4.0 * (cum_runs[hi] - cum_runs[lo])
    if kind == "mix": return (hi - lo) + 8.0 * (cum_runs[hi] - cum_runs[lo]) + 1.0 * (cum_rem[hi] - cum_rem[lo])
    return (hi - lo) + RUN_W * (cum_runs[hi] - cum_runs[lo])


def make_sim(path):
    b = _capi.Binding(ROOT / path)
    b.lib.lobsim_diag_set_order.argtypes = [C.c_void_p, C.c_void_p]
    cfg = _capi.make_cfg(N, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000)
    sim = _capi.Sim(b, cfg)
    sim.add_stream(stream); sim.set_portfolio(0, 10**12)
    sim.reset(starts, 0)
    return b, sim


def orders_for(w, simd_blocks):
    """simd_blocks: [n_simd][per] block numbers of each SIMD (measured placement)."""
    n_simd, per = simd_blocks.shape
    by_w = np.argsort(-w, kind="stable")  # heaviest first
    out = {}
    out["identity"] = np.arange(N, dtype=np.int32)
    bal = np.empty(N, dtype=np.int32)
    for q in range(per):  # quartile q of the sorted envs, dealt to the SIMDs forwards / backwards in turn
        chunk = by_w[q * n_simd:(q + 1) * n_simd]
        if q % 2: chunk = chunk[::-1]
        bal[simd_blocks[:, q]] = chunk
    out["balanced"] = bal
    pk = np.empty(N, dtype=np.int32)
    pk[simd_blocks.reshape(-1)] = by_w  # SIMD 0 gets the four heaviest ...
    out["packed"] = pk
    out["lpt"] = by_w.astype(np.int32)
    # LPT inside residue classes mod 16: env e only ever takes a block b = e mod 16 (it keeps its XCD, b mod 8, launch after launch: its
    # state stays in that XCD's L2) - what 16 sorter waves at the head of the launch could produce, each ranking its own 1/16 of the envs
    cl = np.empty(N, dtype=np.int32)
    for j in range(16):
        members = np.arange(j, N, 16)
        cl[j + 16 * np.arange(len(members))] = members[np.argsort(-w[members], kind="stable")]
    out["lpt16"] = cl
    # what lpt_sort builds since the classes are rotated against the env index (member r of class j = env 16 r + ((j - r) mod 16))
    rot = np.empty(N, dtype=np.int32)
    for j in range(16):
        r = np.arange((N - j + 15) // 16)
        members = 16 * r + np.where(r < N // 16, (j - r) & 15, j)
        rot[j + 16 * r] = members[np.argsort(-w[members], kind="stable")]
    out["lpt16rot"] = rot
    return out


# ---- pass 1: the stamp build (device clock): placement, launch length per policy
b, sim = make_sim(os.environ["PROF_LIB"])
b.lib.lobsim_prof_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros((N, 8), dtype=np.uint64)


def launch(t):
    sim.step_device(acts[t % K].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    b.lib.lobsim_prof_read(sim.h, out.ctypes.data, 1)
    t0, t1 = out[:, 0].astype(np.int64), out[:, 1].astype(np.int64)
    hw = out[:, 2]
    hwid, xcc = (hw & np.uint64(0xffffffff)).astype(np.int64), (hw >> np.uint64(32)).astype(np.int64) & 0xf
    simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
    place = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
    return (t1.max() - t0.min()) * 0.01, (t1 - t0) * 0.01, place, out[:, 3].astype(np.float64)


places = []
for t in range(12):  # identity order: block b = env b
    _, _, place, _ = launch(t)
    places.append(place)
same = [float((places[k] == places[k + 1]).mean()) for k in range(len(places) - 1)]
print(f"{N} envs; block -> SIMD placement equal between consecutive launches (share of blocks): {np.round(same, 3).tolist()}")
place = places[-1]
ids, counts = np.unique(place, return_counts=True)
print(f"  SIMDs used {len(ids)}, blocks per SIMD min {counts.min()} max {counts.max()}")
per = int(counts.min())
simd_blocks = np.stack([np.flatnonzero(place == s)[:per] for s in ids])  # (a SIMD with more blocks than the rest: the extra ones keep their env)
POLS = ["none", "identity", "packed", "lpt", "lpt16", "lpt16:count", "lpt16:runs16", "lpt16:runs4", "lpt16:mix", "lpt16rot"]
res = {p: [] for p in POLS}
dur_top = {p: [] for p in res}
t = 12
null = C.c_void_p(0)
for rep in range(2 * len(POLS)):  # (rotated: every policy meets every phase of the second - all envs start on whole seconds, every tenth launch ends one)
    for pol in POLS[rep % len(POLS):] + POLS[:rep % len(POLS)]:
        w = weights(t, pol.split(":")[1] if ":" in pol else "runs")
        if pol == "none": b.lib.lobsim_diag_set_order(sim.h, null)
        else:
            o = orders_for(w, simd_blocks)[pol.split(":")[0]]
            assert np.array_equal(np.sort(o), np.arange(N))
            b.lib.lobsim_diag_set_order(sim.h, o.ctypes.data)
        L, dur, pl, cyc = launch(t)
        res[pol].append(L); dur_top[pol].append(dur.max())
        if pol == "balanced" and rep == 0 and False:
            blk_of_env = np.empty(N, dtype=np.int64); blk_of_env[o] = np.arange(N)
            print(f"  balanced order: the placement assumed holds for {float((pl == place[blk_of_env]).mean()):.3f} of the envs; SIMD weight sums min {np.bincount(pl, weights=w)[ids].min():.0f} max {np.bincount(pl, weights=w)[ids].max():.0f} (identity: {np.bincount(place, weights=w)[ids].min():.0f} .. {np.bincount(place, weights=w)[ids].max():.0f})")
        t += 1
print("stamp build, device clock: launch = first wave start -> last wave end (us), mean over 2 x len(policies) launches each, policy order rotated; longest wave (us)")
for pol in res: print(f"  {pol:9s} launch {np.mean(res[pol]):6.2f} (min {np.min(res[pol]):.2f} max {np.max(res[pol]):.2f})   longest wave {np.mean(dur_top[pol]):6.2f}")
del sim

# ---- pass 2: the product build + the order entry, wall clock per step (sync - launch - sync: the constant overhead is in every policy)
if os.environ.get("FAST_LIB"):
    b, sim = make_sim(os.environ["FAST_LIB"])
    for t in range(12): sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    sim.sync()
    res = {p: [] for p in res}
    t = 12
    for rep in range(4 * len(POLS)):
        for pol in POLS[rep % len(POLS):] + POLS[:rep % len(POLS)]:
            w = weights(t, pol.split(":")[1] if ":" in pol else "runs")
            if pol == "none": b.lib.lobsim_diag_set_order(sim.h, null)
            else:
                o = orders_for(w, simd_blocks)[pol.split(":")[0]]
                b.lib.lobsim_diag_set_order(sim.h, o.ctypes.data)
            sim.sync()
            t0 = time.perf_counter()
            sim.step_device(acts[t % K].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
            sim.sync()
            res[pol].append(1e6 * (time.perf_counter() - t0))
            t += 1
    print("product build + order entry, wall clock of one launch + sync (us): median / mean over 4 x len(policies) launches each, rotated")
    for pol in res: print(f"  {pol:9s} {np.median(res[pol]):6.2f} / {np.mean(res[pol]):6.2f}")
