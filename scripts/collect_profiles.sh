#!/bin/bash
# Collects the round's measurements on the GPU box into gpurun_out/<tag>/ (copied to profiles/ afterwards; tag = r03).
#   bash scripts/collect_profiles.sh [bench|pmc|models]
# rocprofv3 runs from /tmp with TMPDIR=/tmp, counters in their own passes with --kernel-trace only.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${APN_ROUND_TAG:-r03}
O=$R/gpurun_out/$T
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
what=${1:-bench}
B="--no-cpu-baseline --no-secondary"

if [ "$what" = bench ]; then
    python $R/bench.py > $O/${T}_bench_default.json 2> $O/bench_default.err
    python $R/bench.py --steps 20 --warmup 5 > $O/${T}_bench_driver_flags.json 2>> $O/bench_default.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python $R/bench.py --steps 2000 --warmup 200 $B > $O/prof_bench.log 2>&1
    cp $O/prof_bench/b_kernel_stats.csv $O/${T}_bench_default_kernel_stats.csv
    python $R/scripts/steady_stats.py $O/prof_bench/b_kernel_trace.csv sa_prep_stats 20 3 --csv $O/${T}_bench_default_steady_per_replay.csv > $O/${T}_bench_default_steady.txt
    rm -rf $O/prof_bench
    # the same step on the width-generic kernels (deterministic: no float atomics), for comparison
    python $R/bench.py --steps 2000 --warmup 200 --kernels wide $B 2>/dev/null | grep '^{' > $O/${T}_bench_wide.json
    : > $O/${T}_bench_distributions.jsonl
    for d in D1 D2; do for s in 0 1 2 3 4; do
        python $R/bench.py --steps 400 --warmup 40 $B --distribution $d --seed $s 2>/dev/null | grep '^{' >> $O/${T}_bench_distributions.jsonl
    done; done
    # the N>1 code path at world_size 1 (SyncBatchNorm + gradient all-reduce; captured and eager collectives)
    : > $O/${T}_bench_nccl_world1.jsonl
    for f in "" "--sync-bn off"; do
        APN_BENCH_FORCE_DISTRIBUTED=1 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29741 \
            python $R/bench.py --gpus 1 --no-cpu-baseline $f 2>/dev/null | grep '^{' >> $O/${T}_bench_nccl_world1.jsonl
    done
    python $R/scripts/stamp_passes.py > $O/${T}_pass_stamps.txt 2>/dev/null
fi

if [ "$what" = pmc ]; then
    # HBM traffic and SQ counters of the block on the DEFAULT launch structure (hipGraph replay, index stages on the
    # second stream, tile map): counters in their own passes with --kernel-trace only
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_sa -o p -- python $R/bench.py --steps 200 --warmup 40 $B > $O/pmc_$c.log 2>&1
    done
    python $R/scripts/pmc_summary.py $O/${T}_pmc_fetch_write_summary_block.csv $O/pmc_FETCH_SIZE_sa $O/pmc_WRITE_SIZE_sa > /dev/null
    python $R/scripts/make_traffic_json.py $O/${T}_pmc_fetch_write_summary_block.csv $O/${T}_traffic.json > /dev/null
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq_sa -o p -- python $R/bench.py --steps 200 --warmup 40 $B > $O/pmc_sq.log 2>&1
    python $R/scripts/pmc_summary.py $O/${T}_pmc_sq_summary_block.csv $O/pmc_sq_sa > /dev/null
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_sq2_sa -o p -- python $R/bench.py --steps 200 --warmup 40 $B > $O/pmc_sq2.log 2>&1
    python $R/scripts/pmc_summary.py $O/${T}_pmc_sq2_summary_block.csv $O/pmc_sq2_sa > /dev/null
    rm -rf $O/pmc_*_sa
fi

if [ "$what" = models ]; then
    : > $O/${T}_pointnext_bench.jsonl
    for f in "" "--graph" "--fused" "--fused --graph" "--fused --graph --pipeline" "--fused --wide-first --graph"; do
        python $R/scripts/bench_pointnext.py $f 2>/dev/null | grep '^{' >> $O/${T}_pointnext_bench.jsonl
    done
    : > $O/${T}_gan_step_bench.jsonl
    for n in 1024 2048; do
        python $R/scripts/bench_gan_step.py --points $n 2>/dev/null | grep '^{' >> $O/${T}_gan_step_bench.jsonl
        python $R/scripts/bench_gan_step.py --points $n --graph 2>/dev/null | grep '^{' >> $O/${T}_gan_step_bench.jsonl
        # the step as two lanes of one captured graph (GanStep(overlap=True))
        python $R/scripts/bench_gan_step.py --points $n --mode fused --graph --overlap 2>/dev/null | grep '^{' >> $O/${T}_gan_step_bench.jsonl
    done
    python $R/scripts/bench_gan_step.py --mode fused --graph --overlap --stamps 2>/dev/null > $O/${T}_gan_step_stamps_two_lanes.txt
    python $R/scripts/bench_gan_step.py --mode fused --graph --stamps 2>/dev/null > $O/${T}_gan_step_stamps_single_stream.txt
    python $R/scripts/fps_batch_probe.py 2>/dev/null > $O/${T}_fps_batch_probe.txt
    APN_FPS_RECORDS=1 python $R/scripts/fps_batch_probe.py 2>/dev/null > $O/${T}_fps_batch_probe_records.txt
    python $R/scripts/bench_wide.py 2>/dev/null | grep '^{' > $O/${T}_wide_kernels.jsonl
    python $R/scripts/bench_pointwise.py 2>/dev/null | grep '^{' > $O/${T}_pointwise_layers.jsonl
    rocprofv3 --kernel-trace --output-format csv -d $O/prof_pn -o pn -- python $R/scripts/bench_pointnext.py --fused --graph --steps 12 --warmup 6 > $O/prof_pn.log 2>&1
    python $R/scripts/steady_stats.py $O/prof_pn/pn_kernel_trace.csv sa_geo_kernel 1 3 --csv $O/${T}_pointnext_fused_graph_steady.csv > $O/${T}_pointnext_fused_graph_steady.txt
    rocprofv3 --kernel-trace --output-format csv -d $O/prof_gan -o gan -- python $R/scripts/bench_gan_step.py --mode fused --graph --iters 10 --warmup 4 > $O/prof_gan.log 2>&1
    python $R/scripts/steady_stats.py $O/prof_gan/gan_kernel_trace.csv pointset_group_max_kernel 4 3 --csv $O/${T}_gan_step_fused_graph_steady.csv > $O/${T}_gan_step_fused_graph_steady.txt
    rm -rf $O/prof_pn $O/prof_gan
    rocprofv3 --kernel-trace --output-format csv -d $O/prof_gan -o gan -- python $R/scripts/bench_gan_step.py --mode fused --graph --points 2048 --iters 8 --warmup 3 > $O/prof_gan.log 2>&1
    python $R/scripts/steady_stats.py $O/prof_gan/gan_kernel_trace.csv pointset_group_max_kernel 4 3 > $O/${T}_gan_step_2048_steady.txt
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_pw -o p -- python $R/scripts/bench_pointwise.py --layers decode1 --only planes3 --iters 2 > $O/pmc_pw.log 2>&1
    python $R/scripts/pmc_summary.py $O/${T}_pmc_sq_summary_pointwise.csv $O/pmc_pw > /dev/null
    rm -rf $O/prof_gan $O/pmc_pw
fi
ls -la $O | tail -30
