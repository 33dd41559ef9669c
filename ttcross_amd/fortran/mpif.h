! mpif.h -- single-process stand-in for the MPI symbols the reference's drivers use.  Multi-GPU runs are not
! driven through MPI here: bond groups (TTX_NGROUPS / mybonds) and the RCCL transport live inside libttx.so.
      integer MPI_COMM_WORLD
      parameter (MPI_COMM_WORLD=0)
