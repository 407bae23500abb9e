#!/bin/bash
# Round-5 evidence: ONE run on one box with the final sources -- the -m gpu suite, tools/evidence_min.sh (rocprofv3 statistics,
# PMC traffic, the bench lines), the round's A/Bs, the forward-only configuration, the rule build alone, determinism.
set -u
E=gpurun_out/evidence_r5
mkdir -p $E
python3 -m pytest tests -m gpu -x -q > $E/pytest_gpu.txt 2>&1
bash tools/evidence_min.sh > $E/evidence_min.log 2>&1
cp -r gpurun_out/evidence_min/. $E/min/ 2>/dev/null
python3 tools/ab_step.py tune7=0,1 > $E/ab_tile4_split_step.txt 2>&1
python3 tools/ab_step.py BN_FOLD_FUSED=0,1 > $E/ab_bn_fold_fused.txt 2>&1
python3 tools/ab_coarse.py --levels 3,4,5 --policy 1500 > $E/ab_coarse_layers.txt 2>&1
python3 tools/run_configs.py > $E/configs_cfg2_cfg5.json 2> $E/configs.err
CFG2_PREFETCH=1 python3 tools/cfg2_iters.py DEFER_PREFETCH_READBACK=1 DEFER_PREFETCH_READBACK=0 > $E/cfg2_serving_loop.txt 2>&1
python3 tools/cfg2_iters.py BATCH_BUILD=1 BATCH_BUILD=0 > $E/cfg2_forward_alone.txt 2>&1
for c in 1 0; do SCN_BATCH_BUILD=$c python3 tools/rules_alone.py 20; done > $E/rules_alone.txt 2>&1
python3 tools/determinism_full.py > $E/determinism.txt 2>&1
python3 tools/check_scratch.py > $E/check_scratch.txt 2>&1
du -sh $E
