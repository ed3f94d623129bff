# On the GPU box: the 16-row edge forward with more, smaller-lived workgroups than resident slots (GRL_E16_TUNE build: grid cap and chunk
# size from the environment) -- does a finer deal make it robust against the critic's resident workgroups / uneven waves?
cd $GRAFT_REPO_ROOT
export GRL_LIB=$PWD/_variants/lib_tune.so
line() { python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_ms_per_step']; print('%-26s : %8.2f steps/s  %.4f ms/step   edge fwd %.3f ms' % ('$1', d['value'], d['ms_per_step'], k.get('edge_conv_fwd_kernel', 0)))"; }
for r in 1 2; do
  python bench.py --steps 60 --warmup 8 --pool 16 --no-parity-gate 2>/dev/null | grep "^{" | line "cap 768 (default)"
  GRL_E16_BLOCKS=1536 GRL_E16_NPW=7 python bench.py --steps 60 --warmup 8 --pool 16 --no-parity-gate 2>/dev/null | grep "^{" | line "cap 1536 npw 7"
  GRL_E16_BLOCKS=3072 GRL_E16_NPW=3 python bench.py --steps 60 --warmup 8 --pool 16 --no-parity-gate 2>/dev/null | grep "^{" | line "cap 3072 npw 3"
  GRL_E16_BLOCKS=3072 GRL_E16_NPW=1 python bench.py --steps 60 --warmup 8 --pool 16 --no-parity-gate 2>/dev/null | grep "^{" | line "cap 3072 npw 1"
done
