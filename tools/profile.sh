#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + stats of the default bench command, then the
# HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE need separate passes: 4 TCC slots).
# Usage: bash tools/profile.sh <tag>     -> gpurun_out/prof_<tag>/...
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --cpu-frames 0 --no-pmc \
   > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || echo "trace FAILED"
# the headline job alone (no side workloads, no end-to-end pass: they launch the same kernel on other shapes and would
# blur its average): the per-kernel average of THIS table is what `roofline.avg_launch_us` must agree with
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -- python3 bench.py --cpu-frames 0 --no-pmc \
   --no-side --end-to-end 0 --no-pmc > $OUT/bench_headline_under_rocprof.json 2> $OUT/trace_headline.err || echo "headline trace FAILED"
PM="python3 bench.py --cpu-frames 0 --steps 1 --warmup 0 --no-profile-events --no-side --no-pmc"  # the whole 512-frame job: a row is not read in the window that first touches it
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $PM \
   > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || echo "pmc fetch FAILED"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $PM \
   > $OUT/pmc_write.json 2> $OUT/pmc_write.err || echo "pmc write FAILED"
# summarise on the box (the raw traces are hundreds of MB; only gpurun_out/ travels back, 64 MiB at most)
python3 tools/summarize_prof.py $TAG $OUT/summary
find $OUT -name '*_kernel_trace.csv' -delete; find $OUT -name '*_counter_collection.csv' -delete; find $OUT -name '*.db' -delete
find $OUT -name '*_agent_info.csv' -delete
echo profile done
