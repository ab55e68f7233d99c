#!/bin/bash
# Round-end evidence: full bench line + rocprofv3 kernel stats of the same command (GPU box).
#   bash tools/round_profile.sh <tag>     -> gpurun_out/<tag>_bench_line.json, gpurun_out/<tag>_kernel_stats.csv
TAG=${1:-rX}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R && python bench.py --steps 8 --warmup 3 > gpurun_out/${TAG}_bench.log 2>&1
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench_line.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $R/bench.py --steps 8 --warmup 3 > $R/gpurun_out/${TAG}_prof.log 2>&1
f=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
head -8 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
cat $R/gpurun_out/${TAG}_bench_line.json | cut -c1-900
