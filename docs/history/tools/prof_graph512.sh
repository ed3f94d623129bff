cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/profg512 -o g512 -- python3 $GRAFT_REPO_ROOT/bench.py --minibatch 512 --steps 20 --warmup 4 --no-cpu-baseline --no-roofline > /dev/null 2>&1
