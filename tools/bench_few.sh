# decode at 1..4 sequences: multi-token dmv kernels (default) vs the MFMA path (WRK_DMV_TOKENS=1); bash tools/bench_few.sh [model] [extra bench args]
M=${1:-1.5B}; shift
for B in 1 2 3 4; do
  for TK in 4 1; do
    WRK_DMV_TOKENS=$TK python bench.py --no-cpu-baseline --model $M --batch $B --steps 64 --warmup 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'model':'$M','batch':$B,'dmv_tokens':$TK,'ms_per_step':d['ms_per_step'],'tokens_per_s':d['value'],'frac':d['roofline']['frac']}))"
  done
done
