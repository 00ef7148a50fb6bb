#!/bin/bash
# Every kept workload through tools/profile_workload.sh.  usage: tools/profile_all.sh TAG   (≈ 25 GPU-minutes)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
tools/profile_workload.sh $TAG darkcornell > gpurun_out/prof_$TAG.log 2>&1
tools/profile_workload.sh $TAG darkcornell_mis >> gpurun_out/prof_$TAG.log 2>&1
for wl in veachmis pbrtest pbrtest_textured; do tools/profile_workload.sh $TAG $wl --steps 4 >> gpurun_out/prof_$TAG.log 2>&1; done
for wl in deepbvh scatter; do tools/profile_workload.sh $TAG $wl --steps 2 --warmup 1 >> gpurun_out/prof_$TAG.log 2>&1; done
SKIP_PMC=1 tools/profile_workload.sh $TAG furnace --steps 16 >> gpurun_out/prof_$TAG.log 2>&1
ls gpurun_out/profile | wc -l
