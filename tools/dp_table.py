"""One table of the `data_parallel` blocks of a directory of bench lines (tools/first_multigpu_run.sh).
   python tools/dp_table.py gpurun_out/multigpu"""
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/multigpu"
rows = []
for f in sorted(glob.glob(os.path.join(root, "*.json"))):
    tag = os.path.basename(f)[:-5]
    try:
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception:
        rows.append((tag, None))
        continue
    rows.append((tag, d))
print(f"{'run':44s} {'N':>2s} {'steps/s':>9s} {'ms/step':>8s} {'min..max ms':>15s} {'norm.':>8s} {'spread':>6s} {'2 comms':>7s} {'algo/proto':>12s} | "
      "actor all-reduce mean / max ms | critic's six (sum of means) | join ms")
for tag, d in rows:
    if d is None or d.get("value") is None:
        print(f"{tag:44s} -- no line (failed, hung or invalid: see the .err file)")
        continue
    dp = d.get("data_parallel") or {}
    c = dp.get("collectives", {})
    a = c.get("flat_gradient_actor+loss_records", {})
    crit = sum(v["mean_ms"] for k, v in c.items() if k.startswith(("critic_", "flat_gradient_critic", "loss_critic")))
    env = dp.get("collective_env", {})
    mm = d.get("ms_per_step_min_max") or [float("nan")] * 2
    print(f"{tag:44s} {d['n_gpus']:2d} {d['value']:9.1f} {d['ms_per_step']:8.3f} {mm[0]:7.3f}..{mm[1]:7.3f} {(d.get('value_normalised') or float('nan')):8.1f} "
          f"{dp.get('rank_spread_max_over_min', float('nan')):6.3f} {str(dp.get('communicators', {}).get('two_communicators')):>7s} "
          f"{env.get('NCCL_ALGO', '-') + '/' + env.get('NCCL_PROTO', '-'):>12s} | {a.get('mean_ms', float('nan')):.4f} / {a.get('max_ms', float('nan')):.4f} ({a.get('bytes', 0)} B) | "
          f"{crit:.4f} | {dp.get('main_lane_waits_ms', {}).get('join_critic_lane', float('nan')):.4f}")
