# refresh after the LDS-broadcast lottery stream: profiler passes and bench line of d256 exact
R=$GRAFT_REPO_ROOT
cd $R
bash profiles/measure_r03.sh d256 exact > gpurun_out/m_d256_exact.log 2>&1; echo "d256 exact profiled"
python3 bench.py --workload d256 --arith exact --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_d256_exact_g8.json 2> gpurun_out/b4.err; echo d256 exact done
python3 bench.py --workload d256 --arith fast --steps 3 --warmup 1 > gpurun_out/r03_bench_d256_fast_g8.json 2> gpurun_out/b3.err; echo d256 fast done
