# third pass: the exact unit-cut kernels of Ising D/E (profiler passes and bench line of d256 exact), the old kernels beside them
R=$GRAFT_REPO_ROOT
cd $R
bash profiles/measure_r03.sh d256 exact > gpurun_out/m_d256_exact.log 2>&1; echo "d256 exact profiled"
python3 bench.py --workload d256 --arith exact --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_d256_exact_g8.json 2> gpurun_out/b4.err; echo d256 exact done
TTX_DE_CUT=0 python3 bench.py --workload d256 --arith exact --steps 1 --warmup 0 --no-cpu-baseline --no-extras > gpurun_out/r03_bench_d256_exact_nocut_g8.json 2> gpurun_out/b4b.err; echo d256 exact nocut done
python3 bench.py --workload d64 --arith exact --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r03_bench_d64_exact_g8.json 2> gpurun_out/b9.err; echo d64 done
