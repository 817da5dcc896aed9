#!/bin/bash
# Collects the rocprofv3 evidence of one code state on the GPU box (run from the repo root):
#   bash scripts/collect_profiles.sh gpurun_out/profiles_new [parts]
# parts (default: all of them; one gpurun call is limited to 20 minutes, so they can be run apart):
#   stats  kernel stats + trace summary of the bench command, pipelined occupancy
#   seq    strictly sequential step timeline
#   pmc2   separate FETCH_SIZE / WRITE_SIZE counter passes, config 2        pmc4   the same for config 4
#   sq     SQ counters (wait / active / LDS conflicts) of the three chip-filling kernels, two passes
#   cal    FETCH_SIZE on access patterns with a known number of distinct lines (scripts/diag_membw.py)
#   bench  the bench lines of the same build
#   n2     two ranks on the box's one GPU over gloo (BFF_REHEARSE_ON_ONE_GPU=1): the N > 1 code path of bench.py
# Counter passes collect only this library's kernels (--kernel-include-regex): the synthetic scene generator issues
# ~10^5 torch kernels before the timed region.  Every pass is summarised as soon as it ends; raw traces stay in /tmp.
set -o pipefail
OUT=${1:-gpurun_out/profiles_new}
PARTS=${2:-"stats seq pmc2 pmc4 cal bench"}
mkdir -p "$OUT"
export TMPDIR=/tmp
RAW=/tmp/bff_prof
rm -rf $RAW $RAW.*
B="python3 bench.py --no-cpu-baseline --no-host-inclusive"
INC="--kernel-include-regex bff|gather_stride"
run() { local tag=$1; shift; timeout -k 10 500 "$@" > $RAW.$tag.log 2>&1 || { echo "FAILED: $tag"; tail -5 $RAW.$tag.log; exit 1; }; echo "done: $tag ($SECONDS s)"; }
summarize() { python3 scripts/prof_summarize.py $RAW "$OUT" ${1:-bff} > /dev/null && rm -rf $RAW; }
for part in $PARTS; do case $part in
stats)
    run stats rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -- $B --steps 12 --warmup 4
    grep '^{"metric"' $RAW.stats.log | tail -1 > "$OUT/bench_c2_under_rocprof.json"
    python3 scripts/pipeline_occupancy.py $RAW/stats 8 > "$OUT/pipeline_occupancy.txt"
    summarize ;;
seq)
    run seq rocprofv3 --kernel-trace --output-format csv -d $RAW/seq -- $B --steps 8 --warmup 4 --no-pipeline
    python3 scripts/step_timeline.py $RAW/seq 4 18 > "$OUT/step_timeline_no_pipeline.txt"      # steps 18-21: inside the timed loop
    rm -rf $RAW ;;
pmc2)
    run pmc_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace $INC --output-format csv -d $RAW/pmc_fetch -- $B --steps 8 --warmup 4 --no-pipeline
    summarize
    run pmc_write rocprofv3 --pmc WRITE_SIZE --kernel-trace $INC --output-format csv -d $RAW/pmc_write -- $B --steps 8 --warmup 4 --no-pipeline
    summarize ;;
pmc4)
    run pmc_fetch_c4 rocprofv3 --pmc FETCH_SIZE --kernel-trace $INC --output-format csv -d $RAW/pmc_fetch_c4 -- $B --shape c4 --scenes 2 --steps 4 --warmup 2 --no-pipeline
    summarize
    run pmc_write_c4 rocprofv3 --pmc WRITE_SIZE --kernel-trace $INC --output-format csv -d $RAW/pmc_write_c4 -- $B --shape c4 --scenes 2 --steps 4 --warmup 2 --no-pipeline
    summarize ;;
sq)
    # where the waves' cycles go in the three chip-filling kernels (quad-cycles, MI355X_MICROARCH.md): parked on
    # s_waitcnt / barriers, issue stalls, active; LDS array cycles and the extra cycles of bank conflicts
    K3="--kernel-include-regex merge_components_kernel|project_views_kernel|rle_to_maskbits_kernel"
    run sq1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace $K3 --output-format csv -d $RAW/sq1 -- $B --steps 8 --warmup 4 --no-pipeline
    summarize
    run sq2 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SALU --kernel-trace $K3 --output-format csv -d $RAW/sq2 -- $B --steps 8 --warmup 4 --no-pipeline
    summarize ;;
cal)
    run cal_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace $INC --output-format csv -d $RAW/cal_fetch -- python3 scripts/diag_membw.py
    cp $RAW.cal_fetch.log "$OUT/gather_calibration_stdout.txt"
    summarize gather_stride ;;
bench)
    for s in c2 c1 c5; do timeout -k 10 300 $B --shape $s 2>/dev/null | tail -1 > "$OUT/bench_$s.json"; done
    timeout -k 10 400 $B --shape c4 --scenes 2 --steps 10 --warmup 2 2>/dev/null | tail -1 > "$OUT/bench_c4.json"
    timeout -k 10 300 $B --no-pipeline 2>/dev/null | tail -1 > "$OUT/bench_c2_no_pipeline.json"
    timeout -k 10 400 $B --shape c4 --scenes 2 --steps 5 --warmup 1 --no-pipeline 2>/dev/null | tail -1 > "$OUT/bench_c4_no_pipeline.json"
    for k in 1 2 3; do timeout -k 10 600 python3 bench.py 2>/dev/null | tail -1 > "$OUT/bench_c2_default_run$k.json"; done     # the driver's command: host_inclusive + cpu_baseline
    timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/bench_c2_driver_style.json"
    echo "done: bench ($SECONDS s)" ;;
n2)
    BFF_REHEARSE_ON_ONE_GPU=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29533 bench.py --gpus 2 --steps 12 --warmup 2 --no-cpu-baseline 2> "$OUT/bench_n2_rehearse.err" | grep '^{"metric"' | tail -1 > "$OUT/bench_n2_rehearse.json"
    echo "done: n2 ($SECONDS s)" ;;
esac; done
ls -la "$OUT"
