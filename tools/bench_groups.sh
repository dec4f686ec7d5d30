# concurrent decode pipelines sweep: bash tools/bench_groups.sh [model]
M=${1:-1.5B}
for cfg in "2 1" "2 2" "4 1" "4 2" "4 4" "8 1" "8 2" "8 4" "8 8" "16 1" "16 2" "16 4" "16 8" "32 1" "32 2" "32 4" "32 8" "64 1" "64 2" "64 4" "64 8"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --model $M --batch $1 --groups $2 --steps 48 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'model':'$M','batch':$1,'pipelines':$2,'ms_per_step':d['ms_per_step'],'tokens_per_s':d['value'],'frac':d['roofline']['frac']}))"
done
