# kernel-trace stats + PMC passes of the filter-gradient kernels on tools/wgrad_lab.py (lab): per kernel symbol and grid
#   gpurun -- bash tools/lab/wgrad_probe.sh "wgrad_patch=0;wgrad_patch=1" L6cv2,L8cv2,L10cv2
mkdir -p gpurun_out
ARMS=${1:-"wgrad_patch=0;wgrad_patch=1"}
SHAPES=${2:-L6cv2,L8cv2,L10cv2}
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/wg_kt -o kt -- python $R/tools/wgrad_lab.py --arms "$ARMS" --shapes $SHAPES --rounds 3 --reps 10 > /tmp/wg_kt.log 2>&1
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE" "FETCH_SIZE TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/wg_$i -o pmc -- python $R/tools/wgrad_lab.py --arms "$ARMS" --shapes $SHAPES --rounds 2 --reps 6 > /tmp/wg_$i.log 2>&1
done
cd $R
python tools/kstats.py /tmp/wg_kt "filter-gradient lab ($ARMS; $SHAPES)" 12 > gpurun_out/wgrad_probe_kstats.md 2>&1
python - <<'PY' > gpurun_out/wgrad_probe_counters.txt 2>&1
import glob, sqlite3, re, collections
out = collections.defaultdict(dict)
for d in sorted(glob.glob('/tmp/wg_[0-9]*/')):
    dbs = glob.glob(d + '**/*.db', recursive=True)
    if not dbs: continue
    db = sqlite3.connect(dbs[0])
    for k, g, c, n, a, dur in db.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%wgrad%' group by kernel_name, grid_size, counter_name"):
        key = (re.sub(r'\(anonymous namespace\)::', '', k)[:40], g)
        out[key][c] = a
        out[key].setdefault('_us', {})[d] = dur / 1e3
for key, v in sorted(out.items()):
    print(key, {d[-2:-1]: round(u, 1) for d, u in v['_us'].items()}, 'us per pass')
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    for c, a in sorted(v.items()):
        if c != '_us': print(f"    {c:32s} {a:18.0f}  {a / wc:8.3f} of SQ_WAVE_CYCLES")
PY
cat gpurun_out/wgrad_probe_kstats.md; cat gpurun_out/wgrad_probe_counters.txt
