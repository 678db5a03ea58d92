"""A markdown table of what a profiles/<round>/ directory holds: one row per committed bench line (value, launch time by HIP events, roofline
fraction), the PMC means of its kernel (traffic, L2 hit rate, instruction mix, wait share) and the slowest / mean wave of the stamp build.
usage: python tools/summarize_profiles.py profiles/r6      (build container or GPU box: reads files only)"""
import json, re, sys
from pathlib import Path

d = Path(sys.argv[1])


def pmc(name):
    f = d / name
    if not f.exists():
        return {}
    rows = {}
    for ln in f.read_text().splitlines():
        if ln.count(",") >= 3 and not ln.startswith(("#", "pass_")):
            p = ln.split(",")
            rows[p[1]] = float(p[2])
    return rows


def tail(name):
    f = d / name
    if not f.exists():
        return None
    m = re.search(r"wave cycles\s+mean (\d+).*?MAX (\d+)", f.read_text())
    return (int(m.group(1)), int(m.group(2))) if m else None


lines = [("random", "bench_random.json", "pmc_step_kernel.csv", "tail_profile.txt"), ("random, driver form (--steps 20 --warmup 5)", "bench_random_20steps.json", None, None),
         ("random --agent-orders 128", "bench_random_agent128.json", None, "tail_profile_agent128.txt"), ("teradactyl", "bench_teradactyl.json", "pmc_step_kernel_teradactyl.csv", None),
         ("ad", "bench_ad.json", "pmc_step_kernel_ad.csv", None), ("deep50", "bench_deep50.json", "pmc_step_kernel_deep50.csv", "tail_profile_deep50.txt"),
         ("deep50 --agent-orders 128", "bench_deep50_agent128.json", None, None), ("rs", "bench_rs.json", "pmc_step_kernel_rs.csv", "tail_profile_rs.txt"),
         ("norm", "bench_norm.json", "pmc_step_kernel_norm.csv", None), ("--generic-kernels", "bench_generic.json", "pmc_step_kernel_generic.csv", "tail_profile_generic.txt"),
         ("random --envs 8192", "bench_random_8192envs.json", None, None), ("random --envs 16384", "bench_random_16384envs.json", None, None)]
print("| workload | kernel | `value` M env-steps/s | launch µs (HIP events) | algorithmic MB / launch | `roofline.frac` | FETCH / WRITE MB (PMC) | traffic MB = 2 x FETCH + WRITE | L2 hit | VALU / SALU per env-step | `SQ_WAIT_ANY` share | slowest / mean wave, k cycles |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for name, bj, pc, tp in lines:
    f = d / bj
    if not f.exists():
        continue
    j = json.loads(f.read_text())
    r, n = j["roofline"], j["config"]["envs_per_gpu"]
    c = pmc(pc) if pc else {}
    t = tail(tp) if tp else None
    fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    hit = c.get("TCC_HIT_sum"), c.get("TCC_MISS_sum")
    waves = c.get("SQ_WAVES", n)
    cell = lambda x: "-" if x is None else x
    print(f"| `{name}` | `{r['kernel'].replace('lob_step_kernel', '')}` | {j['value'] / 1e6:.1f} | {r['kernel_ms'] * 1e3:.2f} | {r['algorithmic_bytes_per_env_step'] * n / 1e6:.2f} | {100 * r['frac']:.2f} % | "
          f"{cell(None if fetch is None else f'{fetch * 1024 / 1e6:.2f} / {write * 1024 / 1e6:.2f}')} | {cell(None if fetch is None else f'{(2 * fetch + write) * 1024 / 1e6:.1f}')} | "
          f"{cell(None if hit[0] is None else f'{100 * hit[0] / (hit[0] + hit[1]):.0f} %')} | {cell(None if 'SQ_INSTS_VALU' not in c else '%d / %d' % (c['SQ_INSTS_VALU'] / n, c['SQ_INSTS_SALU'] / n))} | "
          f"{cell(None if 'SQ_WAIT_ANY' not in c else '%.1f %%' % (100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']))} | {cell(None if t is None else f'{t[1] / 1e3:.1f} / {t[0] / 1e3:.1f}')} |")
j = json.loads((d / "bench_random.json").read_text())
for k in ("open_loop_rollout", "cpu_baseline"):
    if k in j:
        print(f"\n`{k}`: {json.dumps(j[k])[:400]}")
if "facade" in j:
    print("\n`facade`: " + json.dumps({k: (round(v / 1e6, 1) if isinstance(v, (int, float)) and v > 1e5 else v) for k, v in j["facade"].items() if not isinstance(v, dict)}))
    if "closed_loop" in j["facade"]:
        print("`closed_loop`: " + json.dumps(j["facade"]["closed_loop"])[:400])
