mkdir -p gpurun_out/r03/pmc_rle; export TMPDIR=/tmp; cd /tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --kernel-include-regex "rle_to_maskbits" --output-format csv -d /tmp/p1 -- python3 $R/scripts/diag_rle.py c2 > $R/gpurun_out/r03/pmc_rle/p1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_SALU --kernel-trace --kernel-include-regex "rle_to_maskbits" --output-format csv -d /tmp/p2 -- python3 $R/scripts/diag_rle.py c2 > $R/gpurun_out/r03/pmc_rle/p2.log 2>&1
for d in p1 p2; do f=$(find /tmp/$d -name "*counter_collection.csv" | head -1); python3 - "$f" <<'PY' > $R/gpurun_out/r03/pmc_rle/$d.txt
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
done
cat $R/gpurun_out/r03/pmc_rle/p1.txt $R/gpurun_out/r03/pmc_rle/p2.txt
