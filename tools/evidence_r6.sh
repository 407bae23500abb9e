#!/bin/bash
# Round-6 evidence: ONE run on one box with the final sources -- rocprofv3 statistics of the bench command, the two PMC traffic
# passes (profiles/r06_traffic.json), clock + MFMA-pipe occupancy inside the step, the bench lines (the default one carries
# parity_fullsize, configs, h2d, cpu_baseline), the assembly loop's A/B and ablations, determinism, scratch, the rule build (sort + hash against occupancy bitmaps: alone, beside a step, by kernel, in the serving loop).
set -u
E=gpurun_out/evidence_r6
mkdir -p $E
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats -- python3 bench.py --no-cpu-baseline --no-configs --host-steps 0 --no-dw-side-stream --steps 5 --warmup 2 > $E/bench_under_rocprof.json 2> $E/stats.err
cp $(find $E/stats -name "*kernel_stats.csv" | head -1) $E/kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats_overlap -- python3 bench.py --no-cpu-baseline --no-configs --host-steps 0 --steps 5 --warmup 2 > $E/bench_under_rocprof_overlap.json 2> $E/stats_overlap.err
cp $(find $E/stats_overlap -name "*kernel_stats.csv" | head -1) $E/kernel_stats_overlap.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $E/fetch -- python3 bench.py --no-cpu-baseline --no-configs --host-steps 0 --no-dw-side-stream --steps 2 --warmup 1 > /dev/null 2> $E/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $E/write -- python3 bench.py --no-cpu-baseline --no-configs --host-steps 0 --no-dw-side-stream --steps 2 --warmup 1 > /dev/null 2> $E/write.err
python3 tools/pmc_traffic.py $E/fetch $E/write $E/traffic.json > $E/traffic.log 2>&1
cp $E/traffic.json profiles/r06_traffic.json
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $E/busy -- python3 bench.py --no-cpu-baseline --no-configs --host-steps 0 --steps 4 --warmup 1 > /dev/null 2> $E/busy.err
python3 tools/pmc_instep.py $E/busy > $E/instep_clock_mfma_busy.txt 2>&1
python3 bench.py > $E/bench.json 2> $E/bench.err
python3 bench.py --no-cpu-baseline --no-configs --steps 20 > $E/bench20.json 2>> $E/bench.err
python3 tools/bench_conv.py --levels 0,1,2 --variants 4 --asm 0,1,2,3,4,5,6,9 > $E/asm_loop_ablation.txt 2>&1
python3 tools/ab_step.py tune28=0,1 > $E/ab_asm_loop_step.txt 2>&1
python3 tools/determinism_full.py > $E/determinism.txt 2>&1
python3 tools/check_scratch.py > $E/check_scratch.txt 2>&1
for o in 0 1 0 1; do SCN_OCC_BUILD=$o python3 tools/rules_alone.py 20 2>&1 | grep "rule build"; done > $E/rules_alone.txt 2>&1
for o in 0 1 0 1; do echo "SCN_OCC_BUILD=$o"; SCN_OCC_BUILD=$o python3 tools/rules_cost_in_step.py 2>/dev/null | head -3; done > $E/rules_cost_in_step.txt 2>&1
SCN_OCC_BUILD=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/rules_occ -- python3 tools/rules_alone.py 10 > /dev/null 2> $E/rules_occ.err
cp $(find $E/rules_occ -name "*kernel_stats.csv" | head -1) $E/rules_occ_kernel_stats.csv
SCN_OCC_BUILD=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/rules_sort -- python3 tools/rules_alone.py 10 > /dev/null 2> $E/rules_sort.err
cp $(find $E/rules_sort -name "*kernel_stats.csv" | head -1) $E/rules_sort_kernel_stats.csv
for o in 0 1; do echo "SCN_OCC_BUILD=$o"; CFG2_PREFETCH=1 SCN_OCC_BUILD=$o python3 tools/cfg2_iters.py 2>/dev/null | tail -6; done > $E/cfg2_serving_loop.txt 2>&1
find $E -name "*.db" -delete
find $E -name "*kernel_trace.csv" -delete
find $E -name "*counter_collection.csv" -delete
du -sh $E
