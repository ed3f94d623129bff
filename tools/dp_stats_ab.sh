# On the GPU box: the data-parallel program (one-rank RCCL group, bench.py --dp-plan) with the advantage statistics reduced once per epoch
# (rollout.publish_advantage_stats) -- run after the DP tests.
cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %5s : %8.2f steps/s  %.4f ms/step  %s' % ('$1', '$2', d['value'], d['ms_per_step'], d['mode']))"; }
for mb in 512 1024 4096; do
  for r in 1 2; do
    python bench.py --dp-plan --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line dp_lanes $mb
    python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | line one_rank $mb
  done
done
python bench.py --dp-plan --minibatch 512 --steps 50 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | grep "^{" | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['data_parallel'], indent=1))"
