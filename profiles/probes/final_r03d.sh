# last pass of round 3: profiler passes and bench lines with the final kernels
R=$GRAFT_REPO_ROOT
cd $R
for w in "d256 exact" "d256 fast" "mvn128 fast" "c64 exact"; do set -- $w; bash profiles/measure_r03.sh $1 $2 > gpurun_out/m_$1_$2.log 2>&1; echo "$1 $2 profiled"; done
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_c64_exact_g8.json 2> gpurun_out/b1.err; echo c64 exact done
python3 bench.py --arith fast --steps 20 --warmup 5 > gpurun_out/r03_bench_c64_fast_g8.json 2> gpurun_out/b2.err; echo c64 fast done
python3 bench.py --workload d256 --arith fast --steps 3 --warmup 1 > gpurun_out/r03_bench_d256_fast_g8.json 2> gpurun_out/b3.err; echo d256 fast done
python3 bench.py --workload d256 --arith exact --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_d256_exact_g8.json 2> gpurun_out/b4.err; echo d256 exact done
python3 bench.py --workload mvn128 --arith fast --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_mvn128_fast_g4.json 2> gpurun_out/b5.err; echo mvn fast done
python3 bench.py --workload d64 --arith exact --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r03_bench_d64_exact_g8.json 2> gpurun_out/b9.err; echo d64 done
