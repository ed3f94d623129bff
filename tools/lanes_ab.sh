cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-22s %5s : %8.2f steps/s  %.4f ms/step  %s' % ('$1', '$2', d['value'], d['ms_per_step'], d['mode']))"; }
for mb in 32 512 4096; do
  for r in 1 2; do
    python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line onegraph $mb
    GRL_LANES=1 python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line lanes $mb
    GRL_LANES=1 GRL_CRITIC_PRIO=0 python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line lanes_noprio $mb
    GRL_OVERLAP_CRITIC=0 python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line serial $mb
  done
done
