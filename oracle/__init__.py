"""CPU oracle for the YOLOv3 detection hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The product path (``yolov3_amd``) never imports this package and fails loudly when the
HIP extension is missing.

Contents
--------
upstream.py     restatement of the un-vendored third-party functions the reference
                delegates to (ultralytics.utils.{metrics,ops,torch_utils}, torchvision.ops.nms).
                Pins: ``ultralytics>=8.4.110`` (reference requirements.txt:18),
                ``torchvision>=0.9.0`` (reference requirements.txt:17).  Exact installed
                versions are unknowable offline -> these are "parity unpinned" upstream; they
                are anchored on the reference's own call sites (cited per function).
yolo_oracle.py  torch-CPU fp32 restatement of the in-repo hot path (model graph walk, Conv /
                Bottleneck / SPP / Concat / Detect decode, non_max_suppression, ComputeLoss),
                each function citing the reference file:line it follows.  Pinned against the
                UNMODIFIED reference (imported through ref_shim.py in the build container) by
                tests/golden/make_golden.py; the resulting vectors live in tests/golden/.
nms_oracle.c    plain-C greedy NMS (torchvision CPU kernel restated), built by oracle/Makefile.
ref_shim.py     stub modules that let the unmodified reference import where /root/reference
                exists (build container only; never on the GPU box).
"""
