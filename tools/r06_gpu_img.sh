#!/bin/bash
# image path: tests, then same-box A/B of the relabelling ahead of the last step
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "image or module or cnn or Cnn or relabel or pipelined or rollout or conv" > gpurun_out/img_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/img_tests.log
tail -6 gpurun_out/img_tests.log | cut -c1-250
for i in 1 2 3; do
  IA_RELABEL_LATE=1 timeout 600 python tools/variant_profile.py image_gail_64x16_cnn 12 2>&1 | tail -1 | cut -c1-100
  timeout 600 python tools/variant_profile.py image_gail_64x16_cnn 12 2>&1 | tail -1 | cut -c1-100
done
