"""CPU-side tests (no GPU): C-ABI surface, host set-up logic, the exact lottery CDF segmentation."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libttx():
    import __graft_entry__ as g
    return ctypes.CDLL(g.build_lib())


def test_cabi_exports_every_declared_symbol(libttx):
    hdr = open(os.path.join(ROOT, "include", "ttx.h")).read()
    names = set(re.findall(r"\b(ttx_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for nm in sorted(names):
        assert hasattr(libttx, nm), f"libttx.so does not export {nm}"


def test_no_gpu_means_loud_failure(libttx):
    """The product has no CPU path: creating an engine without a device must fail with TTX_ENODEV."""
    import torch  # noqa: F401  (only to know whether a GPU is visible)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ttcross_amd import drivers as D
    from ttcross_amd import engine as E
    s = D.ising_setup("c", 6, 33)
    with pytest.raises(E.TTXError, match="no HIP device"):
        E.TTCross(s["n"], s["fun_id"], s["par"], 8, pivoting=2, quad=s["quad"])


def test_bad_arguments_mirror_reference_errors(libttx):
    from ttcross_amd import drivers as D
    from ttcross_amd import engine as E
    s = D.ising_setup("c", 6, 33)
    with pytest.raises(E.TTXError, match="nproc exceeds or equal dimension"):     # lib/dmrgg.f90:114-117
        E.TTCross(s["n"], s["fun_id"], s["par"], 8, pivoting=2, nproc=5)
    with pytest.raises(E.TTXError):
        E.TTCross(s["n"], s["fun_id"], s["par"], 0, pivoting=2)


def test_driver_setup_matches_oracle_setup(oracle_built):
    """ttcross_amd.drivers (product host code) vs the oracle's restatement of the drivers' set-up."""
    from ttcross_amd import drivers as D
    L = ctypes.CDLL(os.path.join(oracle_built, "libttx_oracle.so"))
    dp = ctypes.POINTER(ctypes.c_double)
    L.ttxo_driver_setup.argtypes = [ctypes.c_char, ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, ctypes.POINTER(ctypes.c_int)]
    for kind, m, n in [("c", 6, 33), ("d", 12, 33), ("e", 5, 51), ("s", 4, 33), ("m", 6, 33)]:
        par = np.zeros(2 * n + 1)
        d = m if kind in "sm" else m - 1
        qw = np.zeros(d * n)
        tru, acc, resc = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        L.ttxo_driver_setup(kind.encode(), m, n, par.ctypes.data_as(dp), qw.ctypes.data_as(dp), ctypes.byref(tru), ctypes.byref(acc), ctypes.byref(resc))
        s = D.ising_setup(kind, m, n) if kind in "cde" else D.box_setup({"s": "stdnorm", "m": "mvn"}[kind], m, n)
        assert np.array_equal(s["par"], par[:len(s["par"])])
        assert np.array_equal(np.concatenate(s["quad"]), qw)
        assert s["acc"] == acc.value and bool(s["rescale"]) == bool(resc.value)


def test_share_partition():
    """share() (lib/default.f90:78-97): SURVEY 8(d) quotes own = 1,8,16,24,32,39,47,55,63 for C_64 on 8 ranks."""
    import oracle_lib as O
    own = np.zeros(9, dtype=np.int32)
    O.lib().ttxo_share(1, 62, 8, own.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    assert own.tolist() == [1, 8, 16, 24, 32, 39, 47, 55, 63]


def test_cdf_segments_exact(tmp_path):
    """ttx_cdf.h (shared host/device code of the lottery kernel) vs the plain sequential accumulation of
    lottery2 (lib/rnd.f90:118) for every K in 1..4096, incl. queries at, just below and just above each a_k."""
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "%s/ttcross_amd/csrc/ttx_cdf.h"
int main(){ static ttx_cdfseg seg[TTX_MAXSEG]; long bad=0; srand(1);
 for(int K=1;K<=4096;K++){ int ns=ttx_cdf_build(K,seg); std::vector<double> a(K+1); double c=1.0/K,x=0; a[0]=0; for(int k=1;k<=K;k++){x=x+c;a[k]=x;}
  for(int q=0;q<64;q++){ int k=rand()%%(K+1); double y=(q%%4==0)?a[k]:(q%%4==1)?nextafter(a[k],0.0):(q%%4==2)?nextafter(a[k],2.0):(double)rand()/RAND_MAX; if(y<0)y=0; if(y>=1)y=nextafter(1.0,0.0);
   int want=0; for(int j=K;j>=0;j--) if(a[j]<=y){want=j;break;} if(ttx_cdf_kmax(seg,ns,y)!=want) bad++; } }
 printf("%%ld\n",bad); return bad!=0; }
''' % ROOT)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "0"


# ---- the reference's stream file (lib/ttio.f90): host reader/writer against the genuine reference ----
def _closed_form_cores():
    nn, rr = [3, 4, 2, 5, 3], [1, 2, 3, 2, 2, 1]
    cores = []
    for b in range(1, 6):
        i, j, k = np.meshgrid(np.arange(1, rr[b - 1] + 1), np.arange(1, nn[b - 1] + 1), np.arange(1, rr[b] + 1), indexing="ij")
        cores.append((3 * i + 5 * j + 7 * k + 11 * b) / 16.0)
    return cores


def test_ttio_stream_format_against_reference_file(tmp_path):
    """tests/golden/ttio_5.tt was written by the GENUINE reference's dtt_write (tests/golden/ref_ttio.f90)."""
    from golden_util import GOLDEN
    from ttcross_amd import ttio
    gold = os.path.join(GOLDEN, "ttio_5.tt")
    l, n, r, cores = ttio.read_tt(gold)
    assert l == 1 and list(n) == [3, 4, 2, 5, 3] and list(r) == [1, 2, 3, 2, 2, 1]
    want = _closed_form_cores()
    assert all(np.array_equal(a, b) for a, b in zip(cores, want))
    # the checksum line the reference's dtt_read printed for the same file
    txt = dict(line.split(None, 1) for line in open(os.path.join(GOLDEN, "ttio_5.txt")) if not line.startswith(("write", "read")))
    s = sum(float((c * (np.arange(1, c.shape[0] + 1)[:, None, None] + 2 * np.arange(1, c.shape[1] + 1)[None, :, None]
                        + 3 * np.arange(1, c.shape[2] + 1)[None, None, :] + 4 * b)).sum()) for b, c in enumerate(cores, 1))
    assert s == float(txt["checksum"])
    out = tmp_path / "w.tt"
    ttio.write_tt(out, want)
    assert ttio.same_file(out, gold)
    # error behaviour of dtt_read: missing magic, wrong version (lib/ttio.f90:236-245)
    b = bytearray(open(gold, "rb").read())
    bad = tmp_path / "bad.tt"
    bad.write_bytes(b"XX" + bytes(b[2:]))
    with pytest.raises(ttio.TTFileError, match="not TT header"):
        ttio.read_tt(bad)
    b[8] = 2
    bad.write_bytes(bytes(b))
    with pytest.raises(ttio.TTFileError, match="version"):
        ttio.read_tt(bad)
    bad.write_bytes(open(gold, "rb").read()[:300])
    with pytest.raises(ttio.TTFileError, match="cores"):
        ttio.read_tt(bad)


def test_ttio_written_file_is_read_by_genuine_reference(tmp_path):
    """Where the reference build exists (this container), its own dtt_read must accept a file we wrote."""
    import subprocess
    from golden_util import GOLDEN
    from ttcross_amd import ttio
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_ttio")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_ttio not built (tests/golden/make_golden.sh)")
    mine = tmp_path / "mine.tt"
    ttio.write_tt(mine, _closed_form_cores())
    p = subprocess.run([exe, str(tmp_path / "ref.tt"), str(mine)], capture_output=True, text=True, timeout=60)
    if p.returncode != 0 and "error while loading shared libraries" in p.stderr:
        pytest.skip("reference runtime libraries not present")
    assert p.returncode == 0, p.stderr
    got = [ln for ln in p.stdout.splitlines() if ln.split()[0] in ("lm", "n", "r", "checksum", "read", "write")]
    assert got == open(os.path.join(GOLDEN, "ttio_5.txt")).read().splitlines()
    assert ttio.same_file(tmp_path / "ref.tt", mine)
