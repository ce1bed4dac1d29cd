#!/bin/bash
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_progressive.py tests/test_gpu_parity.py -x -q -m gpu -k "slot or sparse or progress or accumul or rerender or pass" > $O/tests.txt 2>&1
tail -15 $O/tests.txt
python tools/r05_inflight_probe.py 8192 pairs,slots > $O/inflight.json 2> $O/inflight.err
cat $O/inflight.json
