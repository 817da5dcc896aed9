#!/bin/bash
# Collects the rocprofv3 evidence of one code state on the GPU box (run from the repo root):
#   bash scripts/collect_profiles.sh gpurun_out/profiles_new
# kernel stats + trace summary, separate FETCH_SIZE / WRITE_SIZE counter passes, the bench lines of the same
# build, the strictly sequential step timeline and the pipelined occupancy.  Raw traces stay under /tmp.
set -o pipefail
OUT=${1:-gpurun_out/profiles_new}
mkdir -p "$OUT"
export TMPDIR=/tmp
RAW=/tmp/bff_prof
rm -rf $RAW
B="python3 bench.py --no-cpu-baseline"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/stats -- $B --steps 10 --warmup 2 > $RAW.stats.log 2>&1 || { tail -5 $RAW.stats.log; exit 1; }
tail -1 $RAW.stats.log > "$OUT/bench_c2_under_rocprof.json"
python3 scripts/pipeline_occupancy.py $RAW/stats 8 > "$OUT/pipeline_occupancy.txt"
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $RAW/seq -- $B --steps 6 --warmup 2 --no-pipeline > $RAW.seq.log 2>&1 || { tail -5 $RAW.seq.log; exit 1; }
python3 scripts/step_timeline.py $RAW/seq > "$OUT/step_timeline_no_pipeline.txt"
rm -rf $RAW/seq
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $RAW/pmc_fetch -- $B --steps 4 --warmup 1 --no-pipeline > $RAW.f.log 2>&1 || { tail -5 $RAW.f.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $RAW/pmc_write -- $B --steps 4 --warmup 1 --no-pipeline > $RAW.w.log 2>&1 || { tail -5 $RAW.w.log; exit 1; }
python3 scripts/prof_summarize.py $RAW "$OUT"
echo "profiles done"
for s in c2 c1 c4 c5; do timeout -k 10 300 $B --shape $s 2>/dev/null | tail -1 > "$OUT/bench_$s.json"; done
timeout -k 10 300 $B --no-pipeline 2>/dev/null | tail -1 > "$OUT/bench_c2_no_pipeline.json"
timeout -k 10 300 $B --shape c4 --no-pipeline 2>/dev/null | tail -1 > "$OUT/bench_c4_no_pipeline.json"
timeout -k 10 600 python3 bench.py 2>/dev/null | tail -1 > "$OUT/bench_c2_with_cpu_baseline.json"
ls -la "$OUT"
