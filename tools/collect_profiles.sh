#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats of the default bench.py workload,
# then two separate PMC passes (FETCH_SIZE, WRITE_SIZE cannot share a pass).  The PMC passes
# use one stream with the default run's launch size (512 frames per launch): per-dispatch
# counters are only attributable when kernels do not overlap.
# tools/summarize_profiles.py condenses everything into gpurun_out/profile_summary_<tag>/.
# usage: bash tools/collect_profiles.sh <tag>
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu --no-latency --no-configs > $out.stats.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --batch 512 --sub-batches 1 --streams 1 --steps 3 --warmup 1 --no-cpu --no-latency --no-configs > $out.fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --batch 512 --sub-batches 1 --streams 1 --steps 3 --warmup 1 --no-cpu --no-latency --no-configs > $out.write.log 2>&1
python3 tools/summarize_profiles.py $out gpurun_out/profile_summary_$tag
