mkdir -p gpurun_out
for v in ${VARIANTS:-auto v3a v3b v3c v5a v5b}; do
  Y3_CONV=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-layers > gpurun_out/var_$v.log 2>&1
  grep -o '"value": [0-9.]*' gpurun_out/var_$v.log | head -1 | sed "s/^/$v /"
done
python - <<'PY'
import re, glob, os
tabs = {}
for f in sorted(glob.glob('gpurun_out/var_*.log')):
    v = os.path.basename(f)[4:-4]
    for line in open(f):
        m = re.match(r"\s+(L\S+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$", line)
        if m: tabs.setdefault(m.group(1), {})[v] = float(m.group(2))
vs = sorted({v for t in tabs.values() for v in t})
print("layer".ljust(12), " ".join(v.rjust(8) for v in vs))
for l, t in tabs.items():
    best = min(t, key=t.get)
    print(l.ljust(12), " ".join((f"{t.get(v, float('nan')):8.4f}") for v in vs), " best:", best)
PY
