for i in 1 2; do
python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap', d['value'], d['ms_per_step'], d['legs_ms'])"
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sequential', d['value'], d['ms_per_step'], d['legs_ms'])"
done
