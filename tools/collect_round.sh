#!/bin/bash
# Runs on the GPU box (via gpurun): every profile of a round from one tree.  Results land in
# gpurun_out/round_<tag>/ under the names they are committed with in profiles/.
# usage: bash tools/collect_round.sh <tag>        (e.g. r03)
set -e
tag=$1
cd "$GRAFT_REPO_ROOT"
dst=gpurun_out/round_$tag
mkdir -p $dst
bash tools/collect_profiles.sh $tag > $dst/collect.log 2>&1
cp gpurun_out/profile_summary_$tag/kernel_stats.csv $dst/${tag}_kernel_stats.csv
cp gpurun_out/profile_summary_$tag/pmc_traffic.json $dst/${tag}_pmc_traffic.json
bash tools/prof.sh ${tag}one --steps 5 --warmup 2 --no-latency --no-configs --batch 512 --sub-batches 1 --streams 1 > $dst/one_stream.log 2>&1
cp gpurun_out/prof_${tag}one/*/*_kernel_stats.csv $dst/${tag}_kernel_stats_one_stream.csv
bash tools/pmc_sq.sh $tag > $dst/${tag}_sq_counters.txt 2>&1
bash tools/prof_latency.sh > $dst/latency.log 2>&1
cp gpurun_out/prof_lat/*/*_kernel_stats.csv $dst/${tag}_latency_kernel_stats.csv
python3 tools/latency_timeline.py > $dst/${tag}_latency_timeline.txt
for cfg in bf kitti_stereo euroc_stereo euroc_search 4k; do
    bash tools/prof_cfg.sh ${tag}$cfg $cfg > $dst/${tag}_cfg_$cfg.txt 2>&1
    tail -1 gpurun_out/prof_${tag}$cfg.log | cut -c1-1500 >> $dst/${tag}_cfg_$cfg.txt
done
ls $dst
