cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_campaigns.py -m gpu -x -q -k "ising_sweep_bit_exact or bond_groups or variants or multi_process or campaign" 2>&1 | tail -4 || exit 1
bash profiles/probes/de_cut_prof.sh
