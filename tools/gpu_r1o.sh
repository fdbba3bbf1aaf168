mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "stem or model_half or model_rect or config5 or end_to_end" > gpurun_out/o_pytest.log 2>&1; echo "exit $?" >> gpurun_out/o_pytest.log
tail -15 gpurun_out/o_pytest.log
for v in 1 0; do Y3_STEM_PAIR=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); bk=d['roofline']['whole_forward']['by_kernel']; print('pair=$v', d['value'], d['legs_ms'], {k:v['ms'] for k,v in bk.items() if 'stem' in k or 'tc64xtp256>/3x3' in k})"; done
