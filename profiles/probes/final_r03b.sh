# second final pass of round 3 (after k_qr_own and the persistent mvn tables): what changed + the bench lines that cite profiles/
R=$GRAFT_REPO_ROOT
cd $R
bash profiles/measure_r03.sh mvn128 fast > gpurun_out/m_mvn128_fast.log 2>&1; echo "mvn128 fast profiled"
bash profiles/probes/n1_profile.sh > gpurun_out/n1p.log 2>&1; echo "n1 profiled"
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_c64_exact_g8.json 2> gpurun_out/b1.err; echo c64 exact done
python3 bench.py --arith fast --steps 20 --warmup 5 > gpurun_out/r03_bench_c64_fast_g8.json 2> gpurun_out/b2.err; echo c64 fast done
python3 bench.py --workload d256 --arith fast --steps 3 --warmup 1 > gpurun_out/r03_bench_d256_fast_g8.json 2> gpurun_out/b3.err; echo d256 fast done
python3 bench.py --workload d256 --arith exact --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r03_bench_d256_exact_g8.json 2> gpurun_out/b4.err; echo d256 exact done
python3 bench.py --workload mvn128 --arith fast --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_mvn128_fast_g4.json 2> gpurun_out/b5.err; echo mvn fast done
python3 bench.py --workload mvn128 --arith exact --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03_bench_mvn128_exact_g4.json 2> gpurun_out/b6.err; echo mvn exact done
python3 bench.py --workload svd_d64 --steps 5 --warmup 2 > gpurun_out/r03_bench_svd_d64.json 2> gpurun_out/b7.err; python3 bench.py --workload ort_d64 --steps 5 --warmup 2 > gpurun_out/r03_bench_ort_d64.json 2> gpurun_out/b8.err; echo n1 done
