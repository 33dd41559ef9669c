#!/bin/bash
# end of round 2, after k_halfstep_det: full GPU suite, campaigns on the D/E kernels, profile and bench line of D_256
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_gpu_team.log 2>&1; tail -3 gpurun_out/r02_pytest_gpu_team.log
grep -q " passed" gpurun_out/r02_pytest_gpu_team.log || exit 1
grep -q "failed" gpurun_out/r02_pytest_gpu_team.log && exit 1
TTX_FUZZ_CASES=1500 TTX_FUZZ_SEED=97 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -x -k fuzz_random > gpurun_out/r02_team_fuzz1.log 2>&1; tail -2 gpurun_out/r02_team_fuzz1.log
TTX_DE_TEAM_UNITS=1000000 TTX_FUZZ_CASES=700 TTX_FUZZ_SEED=101 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -x -k fuzz_random > gpurun_out/r02_team_fuzz2.log 2>&1; tail -2 gpurun_out/r02_team_fuzz2.log
TTX_DE_TEAM_UNITS=0 TTX_DE_TEAM6_UNITS=1000000 TTX_FUZZ_CASES=700 TTX_FUZZ_SEED=149 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -x -k fuzz_random > gpurun_out/r02_team_fuzz3.log 2>&1; tail -2 gpurun_out/r02_team_fuzz3.log
TTX_SOAK_RUNS=300 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -x -k soak > gpurun_out/r02_team_soak.log 2>&1; tail -2 gpurun_out/r02_team_soak.log
TTX_REFFUZZ_CASES=40 TTX_FUZZ_SEED=103 python -m pytest tests/test_gpu_campaigns.py -m gpu -q -x -k reference_driver > gpurun_out/r02_team_reffuzz.log 2>&1; tail -2 gpurun_out/r02_team_reffuzz.log
timeout -k 10 200 python bench.py --workload d256 --steps 1 --warmup 0 --no-extras > gpurun_out/r02_bench_d256_team.json 2> gpurun_out/r02_bench_d256_team.err
python -c "import json; d=json.load(open('gpurun_out/r02_bench_d256_team.json')); print(d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['avg_launch_us'])"
bash profiles/measure_r02.sh r02_d256team d256 pmc > gpurun_out/r02_measure_team.log 2>&1; tail -25 gpurun_out/r02_measure_team.log
