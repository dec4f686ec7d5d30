# SQ counters of the prefill kernels (own --pmc pass).  usage: bash tools/pmc_kernel.sh <tag>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
TAG=${1:-r03}
WRK_NO_GRAPH=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/prof_c -- python3 $R/tools/prefill_bench.py --batch 32 --prompt 128 --chunk 4096 --repeat 1 > /dev/null 2>&1
F=$(find $O/prof_c -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY' | tee $O/${TAG}_prefill_pmc.txt
import csv, sys, collections
k = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[r["Kernel_Name"]] += 1
for name, c in sorted(k.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:5]:
    print(name[:70], "dispatches", n[name])
    for cn, v in sorted(c.items()): print("   %-28s %.4g" % (cn, v))
PY
rm -rf $O/prof_c
