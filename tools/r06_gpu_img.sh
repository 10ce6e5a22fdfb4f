#!/bin/bash
# image path: split-K linear layer + finer weight-gradient splits: tests, then same-box A/B
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "gemm or cnn or Cnn or image or conv or nature" > gpurun_out/img_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/img_tests.log
tail -6 gpurun_out/img_tests.log | cut -c1-250
for i in 1; do
  IA_IMG_OLD=1 timeout 600 python tools/variant_profile.py image_gail_64x16_cnn 12 2>&1 | tail -1 | cut -c1-100
  timeout 600 python tools/variant_profile.py image_gail_64x16_cnn 12 2>&1 | tail -1 | cut -c1-100
done
