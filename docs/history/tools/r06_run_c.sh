#!/bin/bash
# On the GPU box (round 6, call c): new GPU tests, host enqueue share at shard sizes, box facts, then the whole-millisecond experiment.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(nproc; free -g | head -2) > gpurun_out/box_r06c.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_oneshot.py tests/test_gpu_boundary.py tests/test_gpu_step.py -x -q 2>&1 | tail -8 > gpurun_out/gpu_new_r06c.txt
cat gpurun_out/gpu_new_r06c.txt gpurun_out/box_r06c.txt
for mb in 32 256 512 4096; do
  GRL_BENCH_NO_SELFCHECK=1 python bench.py --minibatch $mb --steps 40 --warmup 8 --pool 16 --no-parity-gate --no-roofline --no-cpu-baseline --repeats 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('%5d frames: %.4f ms/step, host enqueue %.4f ms/step' % ($mb, d['ms_per_step'], d['host_enqueue_ms_per_step']))"
done | tee gpurun_out/host_share_r06c.txt
timeout 1500 bash tools/r06_wholems.sh r06c
