# batched decode sweep (tokens/s, ms/step, fraction of the 8 TB/s HBM roofline): bash tools/bench_batched.sh [model]
M=${1:-1.5B}
for B in 1 2 4 8 16 32 64; do
  python bench.py --no-cpu-baseline --no-prefill --model $M --batch $B --steps 64 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'model':'$M','batch':$B,'ms_per_step':d['ms_per_step'],'tokens_per_s':d['value'],'frac':d['roofline']['frac'],'bytes':d['roofline']['algorithmic_bytes_per_launch']}))"
done
