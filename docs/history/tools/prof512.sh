cd /tmp && export TMPDIR=/tmp
export GRL_B=512 GRL_STEPS=12
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof512 -o p512 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/profile_step.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof512 -name '*kernel_stats.csv' | head -1)
head -40 $f | cut -c1-160
python bench.py --minibatch 512 --no-cpu-baseline --steps 100 2>&1 | tail -1 | cut -c1-400
