#!/bin/bash
# image variant: 3 timed rounds (bench default) against 12, same box
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 600 python tools/variant_profile.py image_gail_64x16_cnn 3 2>&1 | tail -1 | cut -c1-120
  timeout 600 python tools/variant_profile.py image_gail_64x16_cnn 12 2>&1 | tail -1 | cut -c1-120
done
