cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%-12s %5s : %8.2f steps/s  %.4f ms/step' % ('$1', '$2', d['value'], d['ms_per_step']))"; }
for mb in 32 512 4096; do
  for r in 1 2; do
    GRL_ACTOR_PRIO=0 python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line noprio $mb
    python bench.py --minibatch $mb --steps 100 --warmup 8 --pool 16 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line prio $mb
  done
done
for wl in cloth_hepi rigid2_empn; do
  for r in 1 2; do
    GRL_ACTOR_PRIO=0 python bench.py --workload $wl --steps 30 --warmup 4 --pool 8 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line noprio $wl
    python bench.py --workload $wl --steps 30 --warmup 4 --pool 8 --no-parity-gate --no-roofline 2>/dev/null | tail -1 | line prio $wl
  done
done
