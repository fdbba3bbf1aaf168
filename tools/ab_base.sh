# build the library of a committed revision (default HEAD) as yolov3_amd/lib/libyolov3_hip_base.so for a same-box A/B (tools/gpu_ab.sh <that file>);
# delete it afterwards: lab builds must not ship
REV=${1:-HEAD}
rm -rf /tmp/wt && git worktree prune && git worktree add -f /tmp/wt $REV -q && (cd /tmp/wt && python -m yolov3_amd.build > /dev/null) && cp /tmp/wt/yolov3_amd/lib/libyolov3_hip.so yolov3_amd/lib/libyolov3_hip_base.so && git worktree remove --force /tmp/wt && ls -la yolov3_amd/lib/
