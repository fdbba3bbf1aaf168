for v in 256 512 256 512; do
  Y3_CONV_L0=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-layers 2>&1 | grep -E "^\s+L0 |\"value\"" | sed -E 's/.*"value": ([0-9.]+).*/value \1/' | tr '\n' ' '; echo " [L0 tp=$v]"
done
