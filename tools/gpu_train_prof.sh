# kernel-trace of the training step (3 steps + 1 warm-up) -> profiles-ready summaries under gpurun_out/
mkdir -p gpurun_out
R=$PWD
TAG=${1:-t}
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_train -o t -- python $R/tools/train_bench.py --batch 64 --steps 3 --fused > $R/gpurun_out/${TAG}_prof_train.log 2>&1
cd $R
python tools/kstats.py gpurun_out/${TAG}_prof_train "rocprofv3 --kernel-trace: tools/train_bench.py --batch 64 --steps 3 --fused (4 steps incl. warm-up), yolov3 640x640 autocast fp16" > gpurun_out/${TAG}_train_kstats.md 2>&1
python tools/kgroups.py gpurun_out/${TAG}_prof_train 4 gpurun_out/${TAG}_train_groups.json > /dev/null 2>&1
rm -rf gpurun_out/${TAG}_prof_train
head -30 gpurun_out/${TAG}_train_kstats.md
