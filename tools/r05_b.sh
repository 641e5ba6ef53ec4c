#!/bin/bash
# GPU box: the round-5 submit-contract tests and the sweep slice, then timing-only ablations of the vector-memory instructions
T=${1:-r05b}; O=gpurun_out/$T; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_batch.py "tests/test_gpu_parity.py::test_seeded_slice_of_the_randomised_parity_sweep" -x -q -m gpu > $O/gpu_tests.txt 2>&1; rc=$?
tail -15 $O/gpu_tests.txt
shift
tools/r04_abl.sh $T "$@"
