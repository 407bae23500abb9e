#!/bin/bash
# The part of tools/evidence.sh that depends on the kernel sources' hash, for a last-minute kernel change: rocprofv3 stats of
# the bench command, the two PMC traffic passes, and then the default bench line carrying that traffic.
set -u
E=gpurun_out/evidence_min
mkdir -p $E
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats -- python3 bench.py --no-cpu-baseline --no-dw-side-stream --steps 5 --warmup 2 > $E/bench_under_rocprof.json 2> $E/stats.err
cp $(find $E/stats -name "*kernel_stats.csv" | head -1) $E/kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats_overlap -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $E/bench_under_rocprof_overlap.json 2> $E/stats_overlap.err
cp $(find $E/stats_overlap -name "*kernel_stats.csv" | head -1) $E/kernel_stats_overlap.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $E/fetch -- python3 bench.py --no-cpu-baseline --no-dw-side-stream --steps 2 --warmup 1 > /dev/null 2> $E/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $E/write -- python3 bench.py --no-cpu-baseline --no-dw-side-stream --steps 2 --warmup 1 > /dev/null 2> $E/write.err
python3 tools/pmc_traffic.py $E/fetch $E/write $E/traffic.json > $E/traffic.log 2>&1
cp $E/traffic.json profiles/r05_traffic.json
python3 bench.py > $E/bench.json 2> $E/bench.err
python3 bench.py --no-cpu-baseline --steps 20 > $E/bench20.json 2>> $E/bench.err
find $E -name "*.db" -delete
find $E -name "*kernel_trace.csv" -delete
find $E -name "*counter_collection.csv" -delete
du -sh $E
