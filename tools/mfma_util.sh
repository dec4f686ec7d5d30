# MFMA utilisation of the prefill GEMM kernels from hardware counters (own pass: --pmc only).  usage: bash tools/mfma_util.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/prof_m -- python3 $R/tools/prefill_bench.py --batch 32 --prompt 128 --chunk 4096 --repeat 1 > $O/r02_mfma_pmc_bench.json 2>/dev/null
F=$(find $O/prof_m -name "*counter_collection.csv" | head -1)
python3 $R/tools/mfma_util.py $F $O/r02_prefill_mfma_util.json
rm -rf $O/prof_m
