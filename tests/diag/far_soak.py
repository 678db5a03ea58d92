"""Diagnostic (GPU): longer runs of tests/test_gpu_far.py's random engine operations with other seeds, spans and both ladder widths."""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_far as T
for ladder, seed, n_ops, span in ((64, 11, 20000, 900), (128, 12, 20000, 1200), (64, 13, 20000, 400)):
    gpu, cpu = T._engine_pair(ladder, 2048)
    rng = np.random.Generator(np.random.PCG64(seed))
    most = [0, 0]
    n_agent = 0
    for i, op in enumerate(T._random_ops(rng, cpu, n_ops, span)):
        if op[0] == 1 and op[5] == 0 and n_agent > 100: continue  # (128 live agent orders is a capacity of the build, not what this is about)
        fc = cpu.debug_process_orders(0, [op]); fg = gpu.debug_process_orders(0, [op])
        sg, sc = gpu.env_state(0), cpu.env_state(0)
        if sc["status"]:
            print("oracle status", sc["status"], "at op", i, op); break
        assert sg["status"] == 0, (i, op, sg["status"])
        assert list(fg) == list(fc), (i, op)
        if i % 7 == 0:
            for side in (0, 1):
                bg, bc = gpu.book(0, side, 4096), cpu.book(0, side, 4096)
                for a, b in zip(bg, bc): assert np.array_equal(a, b), (i, op, side)
                most[side] = max(most[side], len(bg[0]))
        assert (sg["n_buy_levels"], sg["n_sell_levels"], sg["inventory"], sg["cash"], sg["n_agent_orders"]) == (sc["n_buy_levels"], sc["n_sell_levels"], sc["inventory"], sc["cash"], sc["n_agent_orders"]), (i, op)
        n_agent = sc["n_agent_orders"]
        if i % 2000 == 1999: assert np.array_equal(gpu.queues(0), cpu.queues(0)), i
    assert np.array_equal(gpu.queues(0), cpu.queues(0))
    print("ladder", ladder, "seed", seed, "ops", i + 1, "most levels", most, "ok", flush=True)
    gpu.close()
