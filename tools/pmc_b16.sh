R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() { WRK_NO_GRAPH=1 rocprofv3 --pmc $1 --output-format csv -d $O/prof_x -- python3 $R/bench.py --no-cpu-baseline --batch 16 --steps 6 --warmup 2 > /dev/null 2>&1; F=$(find $O/prof_x -name "*counter_collection.csv" | head -1); python3 - "$F" <<'PY'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:40]; d[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,c in d.items():
    if "gemm" in k or "head" in k or "ln_mix" in k:
        print(k, {a: round(b/max(1,n[(k,a)])) for a,b in c.items()})
PY
rm -rf $O/prof_x; }
run "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_BUSY_CYCLES"
run "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
run "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
