"""Where do device and oracle part on the deep50 workload?  8 sample envs with the test's actions, stepped by two handles
through 300-step rollout launches and by one handle through single steps.  usage: debug_deep.py [lib.so] [chunks]"""
import subprocess, sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import multiprocessing as mp
import bench, oracle_lib as ol
from rl4mm_amd import _capi, _lib
import test_gpu_configs as tg

if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else str(_lib.LIB_PATH)
    chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    missing = [n for n in _capi.PROTOTYPES if ("lobsim_" + n).encode() not in subprocess.run(["nm", "-D", lib], capture_output=True).stdout]
    binding = _capi.Binding(lib, skip=missing)
    with mp.get_context("spawn").Pool(4) as pool:
        streams = pool.map(tg._deep_stream, range(4))
    n, chunk = 8192, 300
    sample = np.array([0, 1, 2, 3, 4100, 4101, 4102, 4103])
    starts, sids = bench.workload_starts("deep50", n, 4, 3)
    m = len(sample)
    cfgm = bench.workload_cfg(_capi, "deep50", m)
    sims = {"rollA": _capi.Sim(binding, cfgm), "rollB": _capi.Sim(binding, cfgm), "single": _capi.Sim(binding, cfgm), "cpu": ol.oracle_sim(cfgm)}
    for sim in sims.values():
        for s in streams: sim.add_stream(s)
        sim.set_portfolio(0, 10**12)
        sim.reset(starts[sample], sids[sample])
    gen = torch.Generator(device="cuda").manual_seed(77)
    D = sims["rollA"].obs_dim
    o, r, d, st = torch.zeros((m, D), dtype=torch.float64).cuda(), torch.zeros(m, dtype=torch.float64).cuda(), torch.zeros(m, dtype=torch.uint8).cuda(), torch.zeros(m, dtype=torch.int32).cuda()
    idx = torch.from_numpy(sample).cuda()
    for c in range(chunks):
        acts = torch.rand((chunk, n, 4), dtype=torch.float64, device="cuda", generator=gen) * 10.0
        am = acts[:, idx].contiguous()
        for k in ("rollA", "rollB"):
            sims[k].rollout_device(_capi.POLICY_GIVEN, [], chunk, am.data_ptr(), 0, o.data_ptr(), 0, r.data_ptr(), d.data_ptr(), 0, st.data_ptr()); sims[k].sync()
        a_host = am.cpu().numpy()
        for t in range(chunk): sims["single"].step(a_host[t])
        ol.oracle_rollout(sims["cpu"], _capi.POLICY_GIVEN, [], chunk, a_host)
        inv = {k: s.portfolios()[0] for k, s in sims.items()}
        ok = {k: bool(np.array_equal(v, inv["cpu"])) for k, v in inv.items() if k != "cpu"}
        lv = max(max(sims["single"].env_state(e)["n_buy_levels"], sims["single"].env_state(e)["n_sell_levels"]) for e in range(m))
        print("chunk", c, ok, "max levels", lv, flush=True)
        if not all(ok.values()):
            for k, v in inv.items(): print("  ", k, v.tolist())
            break
