#!/bin/bash
# On the GPU box: the lane policy on the work-normalised scale (PolicyUpdater._work_frames) -- what it picks for the other workloads at shard sizes,
# against the forms it does not pick.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_policy.txt
: > $OUT
python - >> $OUT 2>&1 <<'PY'
import torch
from geometry_rl_amd import agent, graph, synthetic as syn
import bench
for w in ("rigid_hepi", "cloth_hepi", "rigid2_empn", "rope_hepi_var", "rope_hepi_bf16"):
    spec, cfg, make_obs, name = bench.workload(w)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    actor, critic, proj, loss = agent.build_agent(spec, cfg, device=dev)
    obs = {k: v.to(dev) for k, v in make_obs(64, 100, 0).items()}
    with torch.no_grad():
        actor.forward_diag(*[obs[k] for k in spec.in_features], train=True)
    topo = actor.hyper_data._cache[64]
    e = {"/".join(k): v.n_edges / 64 for k, v in topo["edges"].items()}
    print(f"{w:16s} edges per frame {sum(e.values()):8.1f}  {e}")
PY
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %-22s %5s : %8.2f steps/s  %.4f ms/step  %s' % ('$1', '$2', '$3', d['value'], d['ms_per_step'], d.get('mode','')[:70]))"; }
run() { w=$1; name=$2; mb=$3; shift 3
  python bench.py --workload $w --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline "$@" 2>/dev/null | grep "^{" | tail -1 | line $w $name $mb >> $OUT
}
for r in 1 2; do
  for w in cloth_hepi rigid2_empn rope_hepi_bf16; do
    for mb in 128 512; do
      run $w policy $mb
      GRL_EPOCH_UNROLL_MAX_GATED=0 GRL_EPOCH_GATED_FROM=100000000 run $w ungated_unrolled $mb
      run $w gated_per_step $mb --critic-gate edge0 --unroll 1
      run $w ungated_per_step $mb --no-critic-gate --unroll 1
    done
  done
done
cat $OUT
