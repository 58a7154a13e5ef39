#!/bin/bash
# usage: tools/wgrad_kernel_time.sh <name filter>  -- prints the rocprof average of the wgrad kernels only
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_wgk
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_wgk -- python $GRAFT_REPO_ROOT/tools/wgrad_bench.py 5 "$1" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py stats $(ls gpurun_out/prof_wgk/*/*_results.db | head -1) gpurun_out/wgk.md | grep -E "wgrad_bf16x3|split_tr"
