mkdir -p gpurun_out
run() { timeout 300 python bench.py --mode train --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['final_loss'])"; }
for i in 1 2; do
  Y3_BN_STREAM=0 run "round1-form      "
  Y3_BN_NT_MB=100000 run "grid only        "
  Y3_BN_NT_MB=128 run "nt >= 128 MB     "
  Y3_BN_NT_MB=300 run "nt >= 300 MB     "
  Y3_BN_NT_MB=600 run "nt >= 600 MB     "
  Y3_BN_NT_MB=64 run "nt >= 64 MB      "
done
