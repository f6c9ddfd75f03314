#!/bin/bash
O=gpurun_out/r06s; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -q -m gpu --durations=25 > $O/gpu_suite.log 2>&1
tail -40 $O/gpu_suite.log
