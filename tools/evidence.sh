#!/bin/bash
# Round evidence on the GPU box (run through gpurun; everything lands under gpurun_out/evidence/, the summaries are then
# copied to profiles/ by hand):
#   1. bench.py lines (default, 20 steps)                                    -> bench.json, bench20.json
#   2. timeout 300 rocprofv3 --kernel-trace --stats of the bench command                 -> kernel_stats.csv
#   3. timeout 300 rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)        -> traffic.json (tools/pmc_traffic.py)
#   4. timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES (in-step clock and MFMA-pipe occupancy per kernel)
#   5. PMC groups on the isolated gather-mm layers (tools/pmc_groups.sh)      -> pmc_conv_tile4.txt
#   6. MFMA issue probe, occupancy probe, kernel A/Bs                          -> mfma_probe.txt, occupancy_probe.txt, ab_*.txt
#   7. memory-safety runs of the train step (tools/guard_run.py: guard pages at either end of every tensor)  -> guard_*.txt
#   8. the sort's scratch under the profiler (tools/scratch_fault_repro.sh)      -> scratch_fault_repro.txt
# --pmc runs use --kernel-trace only (the pool refuses PMC together with the hip/hsa trace domains).  Every rocprofv3 run is
# under `timeout`: after a GPU fault the profiler does not exit by itself (round 3, evidence run 3: 40 minutes).
set -u
E=gpurun_out/evidence
mkdir -p $E
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $E/bench.json 2> $E/bench.err
python3 bench.py --no-cpu-baseline --steps 20 > $E/bench20.json 2>> $E/bench.err
# (kernel durations: --no-dw-side-stream keeps dWeight on the main stream in every step, as bench.py does on the steps whose
#  launches carry events, so the averages below are exclusive durations like roofline.avg_launch_ms; the second pass is the
#  default command, in which dWeight overlaps the kernels that follow it and both take longer)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats -- python3 bench.py --no-cpu-baseline --no-dw-side-stream --steps 5 --warmup 2 > $E/bench_under_rocprof.json 2> $E/stats.err
cp $(find $E/stats -name "*kernel_stats.csv" | head -1) $E/kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $E/stats_overlap -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $E/bench_under_rocprof_overlap.json 2> $E/stats_overlap.err
cp $(find $E/stats_overlap -name "*kernel_stats.csv" | head -1) $E/kernel_stats_overlap.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $E/fetch -- python3 bench.py --no-cpu-baseline --no-dw-side-stream --steps 2 --warmup 1 > /dev/null 2> $E/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $E/write -- python3 bench.py --no-cpu-baseline --no-dw-side-stream --steps 2 --warmup 1 > /dev/null 2> $E/write.err
python3 tools/pmc_traffic.py $E/fetch $E/write $E/traffic.json > $E/traffic.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $E/clock -- python3 bench.py --no-cpu-baseline --no-dw-side-stream --steps 2 --warmup 1 > /dev/null 2> $E/clock.err
python3 tools/pmc_instep.py $E/clock > $E/instep_clock.txt 2>&1
bash tools/pmc_groups.sh 1 $E/pmc_tile4
python3 tools/pmc_summary.py $E/pmc_tile4 k_conv_tile4 > $E/pmc_conv_tile4.txt 2>&1
python3 tools/pmc_summary.py $E/pmc_tile4 k_dw_tile > $E/pmc_dw_tile.txt 2>&1
# 6. what an instruction beside an f32 MFMA costs; occupancy probe of k_conv_tile4; k_conv_tile6 against k_conv_tile4
python3 tools/mfma_probe.py > $E/mfma_probe.txt 2>&1
python3 tools/bench_conv.py --levels 0,1 --variants 4 --extra-lds 0,2000,15200 > $E/occupancy_probe.txt 2>&1
python3 tools/ab_step.py DW_SIDE_STREAM=0,1 > $E/ab_dw_side_stream.txt 2>&1
python3 tools/ab_step.py DETERMINISTIC_DW=0,1 > $E/ab_deterministic_dw.txt 2>&1
python3 tools/ab_step.py STREAM_K=0,1 > $E/ab_stream_k_step.txt 2>&1   # (off against the default: the forward launches)
python3 tools/ab_dw_det.py > $E/ab_dw_det_layers.txt 2>&1
for c in 0 1 0 1; do SCN_GRID_CHAIN=$c python3 tools/rules_alone.py 20; done > $E/rules_alone.txt 2>&1
python3 tools/ab_step.py SPLIT_FUSED=0,1 > $E/ab_split_fused.txt 2>&1
python3 tools/ab_step.py tune21=0,1 > $E/ab_expand_persistent_step.txt 2>&1
python3 tools/ab_stream_k.py --levels 0,1 --sk 0,1,0,1 > $E/ab_stream_k_layers.txt 2>&1
python3 tools/ab_stream_k.py --levels 0,1 --sk 0,1,0,1 --first-scene 1100 --residual >> $E/ab_stream_k_layers.txt 2>&1
python3 tools/ab_expand.py > $E/ab_expand_layers.txt 2>&1
python3 tools/run_configs.py > $E/configs_cfg2_cfg5.json 2> $E/configs.err
for mode in end start; do
  GUARD_MODE=$mode timeout 600 python3 -X faulthandler tools/guard_run.py 2> $E/guard_$mode.err | grep guard_run > $E/guard_$mode.txt
done
GUARD_SYNC=stream timeout 600 python3 -X faulthandler tools/guard_run.py 2> $E/guard_stream.err | grep guard_run > $E/guard_stream.txt
GUARD_POISON=0xEF timeout 600 python3 -X faulthandler tools/guard_run.py 2> $E/guard_poison.err | grep guard_run > $E/guard_poison.txt
python3 tools/check_scratch.py > $E/check_scratch.txt 2>&1
# the default bench line once more, now that the traffic of THESE kernel sources exists (bench.py refuses a traffic file
# measured on other sources): this is the line that goes to profiles/rNN_bench.json
cp $E/traffic.json profiles/r05_traffic.json
python3 bench.py > $E/bench_final.json 2>> $E/bench.err
# keep the merge small: raw traces stay on the box
find $E -name "*.db" -delete
find $E -name "*kernel_trace.csv" -size +2M -delete
find $E -name "*counter_collection.csv" -size +8M -delete
du -sh $E
