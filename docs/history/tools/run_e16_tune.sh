#!/bin/bash
# On the GPU box: the 16-row edge kernels under different grid caps / chunk sizes (diagnostic build -DGRL_E16_TUNE in _variants/lib_tune.so)
cd $GRAFT_REPO_ROOT
ARGS="--steps 30 --warmup 5 --pool 16 --no-parity-gate"
for cfg in ${GRL_E16_CFGS:-768,0 256,0 512,0 1024,0 768,1 768,2 768,16 2048,0 4096,1}; do
  b=${cfg%,*}; n=${cfg#*,}
  GRL_E16_BLOCKS=$b GRL_E16_NPW=$n GRL_LIB=$PWD/_variants/lib_tune.so python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_ms_per_step']
print('blocks $b npw $n'.ljust(22), 'steps/s %7.2f' % d['value'], ' '.join('%s %.3f' % (n.replace('_kernel','').replace('edge_conv_','e_').replace('node_mlp_','m_'), k[n]) for n in ('edge_conv_bwd_w_kernel','edge_conv_bwd_x_kernel','edge_conv_fwd_kernel')))"
done
