#!/bin/bash
# round 6, first GPU call: the data-parallel tests (world 2 / 4 on one GPU, bare bench launch, forced handshake failure),
# the predraw / kernel tests the round touched, then the driver's bench command
O=gpurun_out/r06a
mkdir -p $O
timeout 1500 python -m pytest tests/test_distributed.py -m gpu -x -q --durations=20 > $O/dist.log 2>&1; echo "dist rc=$?" | tee -a $O/rc.txt
timeout 900 python -m pytest tests/test_adversarial_gpu.py -m gpu -x -q -k "round_draws or pipelined" > $O/adv.log 2>&1; echo "adv rc=$?" | tee -a $O/rc.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "word_exchange" > $O/kern.log 2>&1; echo "kern rc=$?" | tee -a $O/rc.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
tail -5 $O/dist.log; tail -3 $O/adv.log; tail -3 $O/kern.log
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r06a/bench.json') if l.startswith('{')][-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','ms_per_step_median','ms_per_step_p10','ms_per_step_p90','value_200')}))
print(json.dumps(d['tail_summary']))
PY
