#!/bin/bash
# Regenerates the golden logs in this directory from the GENUINE reference built by `make -C oracle ref`
# (oracle/_ref/, from /root/reference; amdflang + MPICH + MKL sequential).  Only program OUTPUT is stored:
# the per-sweep lines of lib/dmrgg.f90:971-1008 and the footer of the drivers, with the noisy `time:`
# column blanked.  File name = driver arguments (+ _npP for P MPI ranks).  The _npP logs come from the PATCHED multi-rank build
# (oracle/make_ref_mpi.py re-inserts the right-going boundary exchange that lib/dmrgg.f90 lost; the unpatched fp64
# source aborts on more than one rank), the others from the unmodified sources.
set -e
cd "$(dirname "$0")/../.."
export MKL_THREADING_LAYER=SEQUENTIAL OMP_NUM_THREADS=1
R=oracle/_ref
G=tests/golden
norm() { grep -E 'n_evals|computed value|analytic value|correct digits|\.\.\.with' | sed -E 's/time: [0-9.E+-]+/time: -/; s/completed in +[0-9.E+-]+ sec\./completed/'; }
run1() { local drv=$1; shift; local tag=$(echo "$drv $*" | tr ' ' '_'); $R/test_crs_$drv "$@" | norm > $G/$tag.txt; echo $tag; }
runp() { local np=$1; shift; local tag=$(echo "ising $*" | tr ' ' '_')_np$np; /opt/conda/bin/mpiexec -np $np $R/test_crs_ising_mpi "$@" | norm > $G/$tag.txt; echo $tag; }
run1 ising C 6 33 20 2
run1 ising C 16 51 32 2
run1 ising C 64 51 32 2
run1 ising C 5 17 8 0
run1 ising C 6 33 10 1
run1 ising C 8 25 12 3
run1 ising D 6 33 12 2
run1 ising E 5 33 12 2
run1 ising D 12 33 10 2
run1 stdnorm 4 33 10 2
run1 mvn 6 33 12 2
# BASELINE config 4 at full size (about a minute on 8 threads; the result does not depend on the thread count)
OMP_NUM_THREADS=8 run1 mvn 128 33 50 2
runp 2 C 6 33 20 2
runp 4 C 6 33 20 2
runp 8 C 16 51 32 2
runp 8 C 64 51 32 2
runp 3 D 8 33 10 2
# BASELINE config 5 at full size (about 9 minutes each on 8 cores; the result does not depend on the thread count)
OMP_NUM_THREADS=8 run1 ising D 256 101 64 5
runp 8 D 256 101 64 5
# flang random_number stream (first 64 draws, hex) -- pins the RNG restatement
cat > /tmp/ttx_rng.f90 <<'F'
program rng
 double precision :: d(64)
 integer :: i
 call random_number(d)
 do i=1,64
  write(*,'(z16.16)') d(i)
 end do
end program
F
amdflang -O2 /tmp/ttx_rng.f90 -o /tmp/ttx_rng.exe -Wl,-rpath,/opt/rocm/lib/llvm/lib && /tmp/ttx_rng.exe > $G/flang_rng.txt
# dtt_accchk (lib/dmrgg.f90:1081) of the genuine reference through a small driver of our own
amdflang -O2 -fopenmp -I/opt/conda/include -Ioracle/_ref/mod tests/golden/ref_accchk.f90 oracle/_ref/obj/{zero,nan,trans,default,timef,say,rnd,ptype,ort,lr,mat,quad,tt,dmrgg,mvn_pdf}.o \
  -o oracle/_ref/ref_accchk -L/opt/conda/lib -lmpifort -lmpi -lmkl_rt -Wl,-rpath,/opt/conda/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib 2>/dev/null
oracle/_ref/ref_accchk 6 33 12 2 2000 | grep -E "accchk|pivot" > $G/accchk_C_6_33_12_2_2000.txt
oracle/_ref/ref_accchk 8 25 10 3 5000 | grep -E "accchk|pivot" > $G/accchk_C_8_25_10_3_5000.txt
# tt_lib utilities (ort / svd / norm / dot / tijk) of the genuine reference through a small driver of our own
amdflang -O2 -fopenmp -I/opt/conda/include -Ioracle/_ref/mod tests/golden/ref_ttops.f90 oracle/_ref/obj/{zero,nan,trans,default,timef,say,rnd,ptype,ort,lr,mat,quad,tt,dmrgg,mvn_pdf}.o \
  -o oracle/_ref/ref_ttops -L/opt/conda/lib -lmpifort -lmpi -lmkl_rt -Wl,-rpath,/opt/conda/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib 2>/dev/null
oracle/_ref/ref_ttops 6 33 12 2 | grep -vE "n_evals" > $G/ttops_C_6_33_12_2.txt
oracle/_ref/ref_ttops 10 25 16 2 | grep -vE "n_evals" > $G/ttops_C_10_25_16_2.txt
rm -f *.mod
# ztt_quad (complex-weight quadrature) of the genuine reference
amdflang -O2 -fopenmp -I/opt/conda/include -Ioracle/_ref/mod tests/golden/ref_zquad.f90 oracle/_ref/obj/{zero,nan,trans,default,timef,say,rnd,ptype,ort,lr,mat,quad,tt,dmrgg,mvn_pdf}.o \
  -o oracle/_ref/ref_zquad -L/opt/conda/lib -lmpifort -lmpi -lmkl_rt -Wl,-rpath,/opt/conda/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib 2>/dev/null
oracle/_ref/ref_zquad 6 33 12 2 | grep zquad > $G/zquad_C_6_33_12_2.txt
rm -f *.mod
# dtt_write / dtt_read stream format (lib/ttio.f90) of the genuine reference: a 724-byte file with closed-form cores
amdflang -O2 -Ioracle/_ref/mod -c /root/reference/lib/ttio.f90 -o oracle/_ref/obj/ttio.o && mv ttio_lib.mod oracle/_ref/mod/
amdflang -O2 -fopenmp -I/opt/conda/include -Ioracle/_ref/mod tests/golden/ref_ttio.f90 oracle/_ref/obj/{ttio,zero,nan,trans,default,timef,say,rnd,ptype,ort,lr,mat,quad,tt}.o \
  -o oracle/_ref/ref_ttio -L/opt/conda/lib -lmpifort -lmpi -lmkl_rt -Wl,-rpath,/opt/conda/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib 2>/dev/null
oracle/_ref/ref_ttio $G/ttio_5.tt $G/ttio_5.tt | grep -E "info|lm|^n|^r|checksum" > $G/ttio_5.txt
rm -f *.mod
# pivoting = 0 down to the noise floor (pins that the two fibers of lib/dmrgg.f90:492-513 do not enter amax)
run1 ising C 16 33 24 0
runp 5 C 16 33 24 0
run1 ising E 8 25 10 1
run1 ising D 10 17 8 0
runp 3 E 8 25 10 2
run1 ising C 12 9 40 3
