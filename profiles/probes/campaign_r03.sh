#!/bin/bash
# campaigns after the unit-cut kernels of Ising D/E, the persistent mvn tables and k_qr_own went in (end of round 3); run through gpurun
mkdir -p gpurun_out
run() { echo "== $1"; shift; env "$@" timeout -k 10 330 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -x -k "$KSEL" 2>&1 | tail -2; }
KSEL=fuzz_random run "fuzz 2500 cases seed 207 (default: compact D/E tables, cluster kernel for C)" TTX_FUZZ_CASES=2500 TTX_FUZZ_SEED=207
KSEL=fuzz_random run "fuzz 1200 cases seed 209, D/E lane per element with the unit cut" TTX_DE_LANE=1 TTX_FUZZ_CASES=1200 TTX_FUZZ_SEED=209
KSEL=fuzz_random run "fuzz 1200 cases seed 211, compact tables through the generic kernels" TTX_DE_V2=0 TTX_FUZZ_CASES=1200 TTX_FUZZ_SEED=211
KSEL=fuzz_random run "fuzz 800 cases seed 213, round-2 kernels (TTX_DE_CUT=0)" TTX_DE_CUT=0 TTX_FUZZ_CASES=800 TTX_FUZZ_SEED=213
KSEL=fuzz_random run "fuzz 800 cases seed 215, lottery candidates from the compact tables" TTX_DE_LOT_POINT=0 TTX_FUZZ_CASES=800 TTX_FUZZ_SEED=215
KSEL=soak run "soak 600 runs per configuration" TTX_SOAK_RUNS=600
KSEL=multi_process run "multi-process fuzz 16 jobs seed 227" TTX_MPFUZZ_CASES=16 TTX_FUZZ_SEED=227
KSEL=tt_lib run "tt_lib fuzz 40 trains seed 229 (k_qr_own)" TTX_TTOPSFUZZ_CASES=40 TTX_FUZZ_SEED=229
KSEL=reference_driver run "reference-driver fuzz 60 lines seed 231" TTX_REFFUZZ_CASES=60 TTX_FUZZ_SEED=231
