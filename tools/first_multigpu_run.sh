#!/bin/bash
# First contact with a multi-GPU node: nothing of the N > 1 path has ever run on more than one GPU.  One call does
#   (1) bench.py --gpus 2 / 4 / 8 for rigid HEPi and two-agent EMPN, each under a timeout; a hang or a non-zero exit starts a FRESH child
#       with GRL_DP_ONE_COMM=1 (both lanes on one communicator: the documented fallback of agent.PolicyUpdater._plan_dp);
#   (2) at the largest N: a sweep of NCCL_ALGO x NCCL_PROTO for the step's one bandwidth-relevant collective (0.54 MB actor gradient);
#   (3) one table of every line's `data_parallel` block: gpurun_out/multigpu/table.txt  (tools/dp_table.py).
# Usage (from the repo root, on the node):  bash tools/first_multigpu_run.sh [max_gpus]
NMAX=${1:-8}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/multigpu
mkdir -p $O
T=${GRL_MG_TIMEOUT:-420}

run_line() {   # tag, n, workload, extra env...
  local tag=$1 n=$2 wl=$3; shift 3
  env "$@" timeout $T python bench.py --gpus $n --workload $wl --steps 40 --warmup 8 --no-roofline > $O/$tag.json 2> $O/$tag.err
  local rc=$?
  echo "$tag rc=$rc $(tail -c 300 $O/$tag.json | tr -d '\n' | cut -c1-120)"
  return $rc
}

for wl in rigid_hepi rigid2_empn; do
  for n in 2 4 8; do
    [ $n -le $NMAX ] || continue
    if ! run_line ${wl}_n${n} $n $wl; then
      echo "  -> ${wl} n=$n failed or hung (rc above; $T s limit): retrying in a fresh child with GRL_DP_ONE_COMM=1"
      run_line ${wl}_n${n}_onecomm $n $wl GRL_DP_ONE_COMM=1 || echo "  -> the one-communicator fallback failed as well: see $O/${wl}_n${n}_onecomm.err"
    fi
  done
done

N=$NMAX
for algo in Ring Tree; do
  for proto in LL LL128 Simple; do
    run_line sweep_rigid_hepi_n${N}_${algo}_${proto} $N rigid_hepi NCCL_ALGO=$algo NCCL_PROTO=$proto || true
  done
done
# (4) the one-shot all-reduce over hipIpc-mapped peer buffers for that collective (geometry_rl_amd/oneshot.py; OFF by default, never run on
#     more than one GPU before this call): the line's `flat_gradient_actor+loss_records` lane time against the best of the sweep above decides
#     DESIGN.md section 5's go / no-go
for n in 2 $N; do
  [ $n -le $NMAX ] || continue
  run_line oneshot_rigid_hepi_n${n} $n rigid_hepi GRL_DP_ONESHOT=1 || echo "  -> one-shot all-reduce failed at n=$n: see $O/oneshot_rigid_hepi_n${n}.err"
done
# (5) the critic lane's gate (round 6: on from 1024 frames per rank; measured on a one-rank group only): the same lines without it at 2 and 4 ranks
for n in 2 4; do
  [ $n -le $NMAX ] || continue
  run_line nogate_rigid_hepi_n${n} $n rigid_hepi GRL_DP_GATE_FROM=0 || true
done
python tools/dp_table.py $O > $O/table.txt
cat $O/table.txt
