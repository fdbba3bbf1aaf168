# same box: the previous round's tree (git archive <rev> in _prevtree/, its own library) against this tree -- interleaved rounds of the default bench line (inference +
# NMS and the appended train leg, the driver's flags) and of the batch-64 train step alone.
# prepare once, here (the tree is git-ignored): rm -rf _prevtree && mkdir _prevtree && git archive 44e7c3b | tar -x -C _prevtree && (cd _prevtree && python -c 'import __graft_entry__ as g; g.build()')
#   gpurun -- bash tools/gpu_vs_prev_round.sh r05 r06
P=${1:-r05}; N=${2:-r06}
mkdir -p gpurun_out
[ -d _prevtree ] || { echo '_prevtree/ missing: see the line above'; exit 1; }
out=gpurun_out/${N}_same_box_${P}_vs_${N}.txt
echo "# same box, interleaved; default line: python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-clocks; train: python bench.py --mode train --batch 64 --steps 10 --warmup 4" > $out
line='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get("train") or {}; print(sys.argv[1], "inference + NMS", d["value"], "img/s  forward+decode", d["legs_ms"]["forward+decode"], "ms  dominant-group frac", d["roofline"]["frac"], " | appended train leg", t.get("value"), "img/s", t.get("ms_per_step"), "ms")'
tr='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "train step:", d["value"], "img/s ", d["ms_per_step"], "ms/step")'
for r in 1 2; do
  (cd _prevtree && timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-clocks 2>/dev/null) | python -c "$line" "round $r $P" >> $out
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-clocks 2>/dev/null | python -c "$line" "round $r $N" >> $out
  (cd _prevtree && Y3_NO_EXCHANGE_LEG=1 timeout 300 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null) | python -c "$tr" "round $r $P" >> $out
  Y3_NO_EXCHANGE_LEG=1 timeout 300 python bench.py --mode train --batch 64 --steps 10 --warmup 4 2>/dev/null | python -c "$tr" "round $r $N" >> $out
done
cat $out
