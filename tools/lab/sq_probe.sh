# SQ stall counters of conv_v10.h per form and layer shape (lab): two PMC passes over tools/conv_lab.py, summarised per kernel symbol
mkdir -p gpurun_out
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/sq_$i -o pmc -- python $R/tools/conv_lab.py --rounds 2 --reps 10 --only "L6.cv2,L8.cv2,L10.cv2" --arms "v10_half=0;v10_half=1" > /tmp/sq_$i.log 2>&1
done
cd $R
python - <<'PY' > gpurun_out/v10_sq_counters.txt 2>&1
import glob, sqlite3, re, collections
out = collections.defaultdict(dict)
for d in sorted(glob.glob('/tmp/sq_*/')):
    dbs = glob.glob(d + '**/*.db', recursive=True)
    if not dbs: continue
    db = sqlite3.connect(dbs[0])
    for k, g, c, n, a, dur in db.execute("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%v10%' group by kernel_name, grid_size, counter_name"):
        key = (re.sub(r'.*v10_kernelIDF16_', '', k)[:14], g)
        out[key][c] = a
        out[key]['_us'] = dur / 1e3
for key, v in sorted(out.items()):
    print(key, f"{v['_us']:.1f} us")
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    for c, a in sorted(v.items()):
        if c != '_us': print(f"    {c:32s} {a:16.0f}  {a / wc:8.3f} of SQ_WAVE_CYCLES")
PY
cat gpurun_out/v10_sq_counters.txt
