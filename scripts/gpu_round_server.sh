#!/bin/bash
# one GPU-box visit: server tests, server bench at the headline size
set -x
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_server.py tests/test_gpu_dropin.py -x -q -s ) > gpurun_out/server_tests.log 2>&1
tail -5 gpurun_out/server_tests.log
( time timeout 900 python scripts/server_bench.py --procs ${PROCS:-1,16,64,256,1024} ) > gpurun_out/server_bench.log 2>&1
tail -25 gpurun_out/server_bench.log
