#!/bin/bash
# On the GPU box: bench every _variants/lib_*.so (alternating, two rounds) on this one box; prints steps/s and the five MFMA kernels' ms per step.
cd $GRAFT_REPO_ROOT
export GRL_ALLOW_DIAG_LIB=1   # knock-out variants are GRL_DIAG builds: geometry_rl_amd/hip.py loads them only with this set
ARGS=${GRL_VARIANT_ARGS:---steps 30 --warmup 5 --pool 16 --no-parity-gate}
for round in ${GRL_VARIANT_ROUNDS:-1 2}; do
  for lib in _variants/lib_*.so; do
    GRL_LIB=$PWD/$lib python bench.py $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['per_kernel_ms_per_step']
print('$lib'.split('lib_')[1][:-3].ljust(14), 'steps/s %7.2f' % d['value'], ' '.join('%s %.3f' % (n.replace('_kernel','').replace('edge_conv_','e_').replace('node_mlp_','m_').replace('edge_','e_'), k[n]) for n in ('edge_bwd16_kernel','edge_conv_bwd_w_kernel','node_mlp_bwd16_kernel','node_mlp_bwd_fused_kernel','edge_conv_bwd_x_kernel','edge_conv_fwd_kernel','node_mlp_fwd_kernel') if n in k))"
  done
done
