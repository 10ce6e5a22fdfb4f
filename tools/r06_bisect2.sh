#!/bin/bash
# the same pair (data-parallel tests, then the first packed `update` case) on an older tree (_old) and on this one
K='test_ppo_epochs_match_oracle and update-11-3-32'
for d in _old .; do
  ( cd $d; timeout 900 python -m pytest tests/test_distributed.py tests/test_kernels_gpu.py -m gpu -x -q -k "(test_distributed) or ($K)" -p no:cacheprovider > /tmp/bis_$$.txt 2>&1; echo "$d: rc=$? $(tail -1 /tmp/bis_$$.txt | cut -c1-100)" )
done
