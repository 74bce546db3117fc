#!/bin/bash
# Collects the round's measurements on the GPU box into gpurun_out/r02/ (copied to profiles/ afterwards).
#   bash scripts/collect_profiles.sh [bench|pmc|models]
# rocprofv3 runs from /tmp with TMPDIR=/tmp, counters in their own passes with --kernel-trace only.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
what=${1:-bench}
B="--no-cpu-baseline --no-secondary"

if [ "$what" = bench ]; then
    python $R/bench.py > $O/r02_bench_default.json 2> $O/bench_default.err
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python $R/bench.py --steps 200 --warmup 20 $B > $O/prof_bench.log 2>&1
    cp $O/prof_bench/b_kernel_stats.csv $O/r02_bench_default_kernel_stats.csv
    python $R/scripts/steady_stats.py $O/prof_bench/b_kernel_trace.csv sa_prep_features 20 3 --csv $O/r02_bench_default_steady_per_replay.csv > $O/r02_bench_default_steady.txt
    python $R/scripts/chain_gaps.py $O/prof_bench/b_kernel_trace.csv sa_prep_features 40 > $O/r02_bench_default_chain.txt
    rm -rf $O/prof_bench
    # the same step on the width-generic kernels (deterministic: no float atomics), for comparison
    python $R/bench.py --steps 200 --warmup 20 --kernels wide $B 2>/dev/null | grep '^{' > $O/r02_bench_wide.json
    rocprofv3 --kernel-trace --output-format csv -d $O/prof_wide -o b -- python $R/bench.py --steps 200 --warmup 20 --kernels wide $B > $O/prof_wide.log 2>&1
    python $R/scripts/steady_stats.py $O/prof_wide/b_kernel_trace.csv wide_fwd_prep 20 3 > $O/r02_bench_wide_steady.txt
    python $R/scripts/chain_gaps.py $O/prof_wide/b_kernel_trace.csv wide_fwd_prep 40 > $O/r02_bench_wide_chain.txt
    rm -rf $O/prof_wide
    : > $O/r02_bench_distributions.jsonl
    for d in D1 D2; do for s in 0 1 2 3 4; do
        python $R/bench.py --steps 100 --warmup 20 $B --distribution $d --seed $s 2>/dev/null | grep '^{' >> $O/r02_bench_distributions.jsonl
    done; done
fi

if [ "$what" = pmc ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_sa -o p -- python $R/bench.py --steps 20 --warmup 5 --graph off --pipeline off $B > $O/pmc_$c.log 2>&1
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_pn -o p -- python $R/scripts/bench_pointnext.py --fused --steps 5 --warmup 2 >> $O/pmc_$c.log 2>&1
    done
    python $R/scripts/pmc_summary.py $O/r02_pmc_fetch_write_summary_block.csv $O/pmc_FETCH_SIZE_sa $O/pmc_WRITE_SIZE_sa > /dev/null
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${c}_wide -o p -- python $R/bench.py --steps 20 --warmup 5 --graph off --pipeline off --kernels wide $B >> $O/pmc_$c.log 2>&1
    done
    python $R/scripts/pmc_summary.py $O/r02_pmc_fetch_write_summary_block_wide.csv $O/pmc_FETCH_SIZE_wide $O/pmc_WRITE_SIZE_wide > /dev/null
    rm -rf $O/pmc_*_wide
    python $R/scripts/pmc_summary.py $O/r02_pmc_fetch_write_summary_classifier.csv $O/pmc_FETCH_SIZE_pn $O/pmc_WRITE_SIZE_pn > /dev/null
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_sq_pn -o p -- python $R/scripts/bench_pointnext.py --fused --steps 5 --warmup 2 > $O/pmc_sq.log 2>&1
    python $R/scripts/pmc_summary.py $O/r02_pmc_sq_summary_classifier.csv $O/pmc_sq_pn > /dev/null
    rm -rf $O/pmc_*_sa $O/pmc_*_pn $O/pmc_sq_pn
fi

if [ "$what" = models ]; then
    : > $O/r02_pointnext_bench.jsonl
    for f in "" "--graph" "--fused" "--fused --graph" "--fused --graph --pipeline" "--fused --wide-first --graph"; do
        python $R/scripts/bench_pointnext.py $f 2>/dev/null | grep '^{' >> $O/r02_pointnext_bench.jsonl
    done
    : > $O/r02_gan_step_bench.jsonl
    for n in 1024 2048; do
        python $R/scripts/bench_gan_step.py --points $n 2>/dev/null | grep '^{' >> $O/r02_gan_step_bench.jsonl
        python $R/scripts/bench_gan_step.py --points $n --graph 2>/dev/null | grep '^{' >> $O/r02_gan_step_bench.jsonl
    done
    python $R/scripts/bench_wide.py 2>/dev/null | grep '^{' > $O/r02_wide_kernels.jsonl
    python $R/scripts/bench_pointwise.py 2>/dev/null | grep '^{' > $O/r02_pointwise_layers.jsonl
    rocprofv3 --kernel-trace --output-format csv -d $O/prof_pn -o pn -- python $R/scripts/bench_pointnext.py --fused --graph --steps 12 --warmup 6 > $O/prof_pn.log 2>&1
    python $R/scripts/steady_stats.py $O/prof_pn/pn_kernel_trace.csv fps_ 4 3 --csv $O/r02_pointnext_fused_graph_steady.csv > $O/r02_pointnext_fused_graph_steady.txt
    rocprofv3 --kernel-trace --output-format csv -d $O/prof_gan -o gan -- python $R/scripts/bench_gan_step.py --mode fused --graph --iters 10 --warmup 4 > $O/prof_gan.log 2>&1
    python $R/scripts/steady_stats.py $O/prof_gan/gan_kernel_trace.csv pointset_group_max_kernel 4 3 --csv $O/r02_gan_step_fused_graph_steady.csv > $O/r02_gan_step_fused_graph_steady.txt
    rm -rf $O/prof_pn $O/prof_gan
    rocprofv3 --kernel-trace --output-format csv -d $O/prof_gan -o gan -- python $R/scripts/bench_gan_step.py --mode fused --graph --points 2048 --iters 8 --warmup 3 > $O/prof_gan.log 2>&1
    python $R/scripts/steady_stats.py $O/prof_gan/gan_kernel_trace.csv pointset_group_max_kernel 4 3 > $O/r02_gan_step_2048_steady.txt
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc_pw -o p -- python $R/scripts/bench_pointwise.py --layers decode1 --only planes3 --iters 2 > $O/pmc_pw.log 2>&1
    python $R/scripts/pmc_summary.py $O/r02_pmc_sq_summary_pointwise.csv $O/pmc_pw > /dev/null
    rm -rf $O/prof_gan $O/pmc_pw
fi
ls -la $O | tail -30
