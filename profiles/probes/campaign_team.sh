#!/bin/bash
# campaigns after k_halfstep_det went in (end of round 2); run through gpurun from the repo root
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" python -m pytest tests/test_gpu_campaigns.py -m gpu -q -x -k "$KSEL" 2>&1 | tail -2; }
KSEL=fuzz_random run "fuzz 3000 cases seed 107 (default: teams while the ranks are small)" TTX_FUZZ_CASES=3000 TTX_FUZZ_SEED=107
KSEL=fuzz_random run "fuzz 1500 cases seed 109, teams at every rank" TTX_DE_TEAM_UNITS=1000000 TTX_FUZZ_CASES=1500 TTX_FUZZ_SEED=109
KSEL=fuzz_random run "fuzz 800 cases seed 113, teams at every rank, general division" TTX_DE_TEAM_UNITS=1000000 TTX_DE_FASTDIV=0 TTX_FUZZ_CASES=800 TTX_FUZZ_SEED=113
KSEL=soak run "soak 1500 runs per configuration" TTX_SOAK_RUNS=1500
KSEL=multi_process run "multi-process fuzz 24 jobs seed 127" TTX_MPFUZZ_CASES=24 TTX_FUZZ_SEED=127
KSEL=reference_driver run "reference-driver fuzz 120 lines seed 131" TTX_REFFUZZ_CASES=120 TTX_FUZZ_SEED=131
KSEL=host_callback run "host-callback fuzz seed 137" TTX_FUZZ_CASES=300 TTX_FUZZ_SEED=137
