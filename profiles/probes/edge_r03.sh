#!/bin/bash
# edge shapes through the round-3 kernels (Ising D / E with the unit cut, exact and fast; mvn fast with persistent tables), one process per case;
# every case is compared with the oracle by tests elsewhere -- this probe looks for faults, hangs and refusals at the size limits
cd $GRAFT_REPO_ROOT
run() { timeout -k 5 120 python3 profiles/probes/edge_shapes.py "$@" 2>&1 | tail -1; }
for id in 2 3; do
 for ar in exact fast; do
  export TTX_ARITH=$ar
  echo "== ident $id arith $ar"
  run "[5,5]" $id 4 2 1
  run "[5,5,5]" $id 4 2 1
  run "[5,5,5]" $id 4 2 2
  run "[1,1,1,1]" $id 2 2 1
  run "[7,3,5,2,6,4]" $id 5 2 2
  run "[9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9,9]" $id 6 3 4
  run "$(python3 -c 'print([3]*300)')" $id 3 1 8
  run "$(python3 -c 'print([2]*700)')" $id 2 1 4
  run "$(python3 -c 'print([33]*40)')" $id 40 2 3
  run "$(python3 -c 'print([17]*20)')" $id 17 -1 1
  run "$(python3 -c 'print([65]*12)')" $id 100 2 2
 done
done
unset TTX_ARITH
echo "== mvn fast, large d"
timeout -k 5 200 python3 - <<'P' 2>&1 | tail -3
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from ttcross_amd import drivers as D, engine as E
for d, n, r, ng in [(200, 9, 12, 4), (64, 33, 40, 8), (3, 9, 6, 2), (2, 5, 4, 1)]:
    s = D.box_setup("mvn", d, n)
    try:
        tt = E.TTCross(s["n"], s["fun_id"], s["par"], r, pivoting=2, accuracy=s["acc"], quad=s["quad"], tru=s["tru"], aux=s["aux"], nproc=ng, arith="fast").run()
        print("ok mvn", d, n, r, ng, "sweeps", len(tt.sweeps()), "value", tt.quad(s["quad"]), flush=True)
    except E.TTXError as e:
        print("refused mvn", d, n, r, ng, str(e)[:150], flush=True)
P
