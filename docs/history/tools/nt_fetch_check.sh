#!/bin/bash
# On the GPU box: FETCH_SIZE / WRITE_SIZE of node_mlp_bwd_fused_kernel with streaming (nt) loads of x2 / dOut (shipped) against plain loads
# (_variants/lib_nt0.so built with -DGRL_MLP_NT=0).  Separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE needs its own pass).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GRL_STEPS=2
for v in nt nt0; do
  if [ $v = nt0 ]; then export GRL_LIB=$R/_variants/lib_nt0.so; else unset GRL_LIB; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/ntF_$v -- python3 $R/tools/profile_step.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/ntW_$v -- python3 $R/tools/profile_step.py > /dev/null 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for v in ("nt", "nt0"):
    for tag, cname in (("F", "FETCH_SIZE"), ("W", "WRITE_SIZE")):
        f = glob.glob(f"gpurun_out/nt{tag}_{v}/*/*counter_collection.csv")[0]
        acc, n = collections.Counter(), collections.Counter()
        seen = set()
        for r in csv.DictReader(open(f)):
            if "node_mlp_bwd_fused" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                acc[cname] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"]); n[cname] += 1
        mb = acc[cname] * 1024 / max(n[cname], 1) / 1e6 * (2 if tag == "F" else 1)
        print(f"node_mlp_bwd_fused_kernel [{ 'nt loads (shipped)' if v == 'nt' else 'plain loads (-DGRL_MLP_NT=0)'}] {cname}: {mb:8.1f} MB per launch ({'x2 corrected' if tag == 'F' else 'as read'}), {n[cname]} launches")
PY
