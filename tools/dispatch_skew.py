"""Diagnostic: when do the 4096 waves of a step launch start and end, and where do they run?  (-DLOBSIM_PROFILE -DLOBSIM_PROFILE_SKEW
build: s_memrealtime at the top and the bottom of every wave, HW_ID / XCC_ID.)  Answers: how long the dispatcher takes to hand out a
launch, whether the last wave to finish is a late starter or a long runner, how many waves share a SIMD.
usage: PROF_LIB=build/variants/skew.so python tools/dispatch_skew.py [n_envs]"""
import ctypes as C, os, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from rl4mm_amd import _capi

b = _capi.Binding(ROOT / os.environ["PROF_LIB"])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 80
cfg = _capi.make_cfg(N, 10, bench.default_features(_capi), step_us=100_000, episode_us=1_800_000_000)
sim = _capi.Sim(b, cfg)
sim.add_stream(bench.build_stream(60)); sim.set_portfolio(0, 10**12)
starts = (36_000 + np.random.default_rng(1).integers(0, 60, size=N)) * 1_000_000
sim.reset(starts, 0)
dev = torch.device("cuda", 0)
acts = torch.from_numpy(np.random.default_rng(2).uniform(0, 10, size=(K, N, 4))).to(dev)
obs = torch.zeros((N, sim.obs_dim), dtype=torch.float64, device=dev)
reward = torch.zeros(N, dtype=torch.float64, device=dev); done = torch.zeros(N, dtype=torch.uint8, device=dev)
out = np.zeros((N, 8), dtype=np.uint64)
b.lib.lobsim_prof_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
b.lib.lobsim_prof_read_messages.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
sorter = np.zeros((N, 16), dtype=np.uint64)  # (rows N .. of the stamp array: the launch's 16 sorter waves, lpt_sort)
rows, srows = [], []
for t in range(K):
    sim.step_device(acts[t].data_ptr(), 0, obs.data_ptr(), 0, reward.data_ptr(), done.data_ptr(), 0)
    b.lib.lobsim_prof_read(sim.h, out.ctypes.data, 1)
    b.lib.lobsim_prof_read_messages(sim.h, sorter.ctypes.data, 1)
    if t < 20: continue
    t0, t1 = out[:, 0].astype(np.int64), out[:, 1].astype(np.int64)
    hw, cyc = out[:, 2], out[:, 3].astype(np.float64)
    base = t0.min()
    st, en = (t0 - base) * 0.01, (t1 - base) * 0.01  # us (100 MHz)
    last = int(np.argmax(en))
    sw = sorter.reshape(-1, 8)[:16]
    if sw[:, 0].any(): srows.append(((sw[:, 0].astype(np.int64) - base) * 0.01, (sw[:, 1].astype(np.int64) - base) * 0.01, en.max()))
    hwid, xcc = (hw & np.uint64(0xffffffff)).astype(np.int64), (hw >> np.uint64(32)).astype(np.int64) & 0xf
    simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
    place = ((xcc * 8 + se) * 2 + sh) * 16 * 4 + cu * 4 + simd
    per_simd = np.bincount(place, minlength=1)
    per_simd = per_simd[per_simd > 0]
    per_cu = np.bincount(place // 4); per_cu = per_cu[per_cu > 0]
    rows.append(dict(launch=en.max(), start_p50=np.median(st), start_p99=np.percentile(st, 99), start_max=st.max(), last_start=st[last], last_dur=en[last] - st[last],
                     dur_max=(en - st).max(), dur_mean=(en - st).mean(), simd_used=len(per_simd), simd_max=per_simd.max(), cu_used=len(per_cu), cu_max=per_cu.max(), cu_min=per_cu.min(),
                     order_corr=np.corrcoef(np.arange(N), st)[0, 1], xccs=len(np.unique(xcc))))
    # does a wave run longer when its SIMD-mates are heavy?  (duration of the other waves of the same SIMD, for all waves and for the longest 1 %)
    dur = en - st
    simd_sum = np.bincount(place, weights=dur)[place]
    mates = (simd_sum - dur) / np.maximum(np.bincount(place)[place] - 1, 1)
    top = dur >= np.percentile(dur, 99)
    rows[-1].update(mates_all=mates.mean(), mates_top=mates[top].mean(), corr_mates=np.corrcoef(dur, mates)[0, 1], simd_sum_max=simd_sum.max(), simd_sum_mean=simd_sum.mean())
    if t == 20:
        print("first measured launch: start time (us) of env 0, 1, 2, 3, 255, 256, 1023, 1024, 4095:", np.round(st[[0, 1, 2, 3, 255, 256, 1023, 1024, N - 1]], 2).tolist())
        print("  xcc of env 0..15:", xcc[:16].tolist(), " cu:", cu[:16].tolist(), " simd:", simd[:16].tolist())
A = lambda k: np.array([r[k] for r in rows], dtype=np.float64)
print(f"{N} envs, {len(rows)} launches (us): launch (first start -> last end) {A('launch').mean():.2f}; wave start p50 {A('start_p50').mean():.2f} p99 {A('start_p99').mean():.2f} max {A('start_max').mean():.2f}")
print(f"  the wave that ends last: starts at {A('last_start').mean():.2f}, runs {A('last_dur').mean():.2f}; longest wave {A('dur_max').mean():.2f}, mean wave {A('dur_mean').mean():.2f}")
print(f"  placement: {A('xccs').mean():.0f} XCCs, {A('cu_used').mean():.0f} CUs used, waves per CU {A('cu_min').mean():.1f} .. {A('cu_max').mean():.1f}, SIMDs used {A('simd_used').mean():.0f}, most waves on one SIMD {A('simd_max').mean():.1f}; corr(env index, start time) {A('order_corr').mean():.2f}")
print(f"  SIMD-mates: mean duration of the other waves on a wave's SIMD {A('mates_all').mean():.2f} us over all waves, {A('mates_top').mean():.2f} us for the longest 1 % of waves; corr(duration, mates' mean duration) {A('corr_mates').mean():.2f}; wave-time per SIMD mean {A('simd_sum_mean').mean():.1f} us, busiest SIMD {A('simd_sum_max').mean():.1f} us")
if srows:
    s0, s1, le = np.array([r[0] for r in srows]), np.array([r[1] for r in srows]), np.array([r[2] for r in srows])
    print(f"  sorter waves (16 at the head of the launch): start {s0.mean():.2f} us after the first env wave (min {s0.min(axis=1).mean():.2f}), run {(s1 - s0).mean():.2f} us (longest of a launch {(s1 - s0).max(axis=1).mean():.2f}), "
          f"the last one ends at {s1.max(axis=1).mean():.2f} us - the launch's last env wave at {le.mean():.2f} us")
