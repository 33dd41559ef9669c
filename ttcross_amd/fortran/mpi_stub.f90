! MPI stand-in for the symbols the reference's drivers use (see mpif.h).  The MI355X engine does not run over MPI:
! a multi-GPU job is one process per GPU started by any launcher that exports TTX_WORLD_RANK / TTX_WORLD_SIZE
! (+ TTX_COMM_FILE for the RCCL bootstrap, TTX_DEVICE for the GPU ordinal); mpi_comm_rank / mpi_comm_size report
! exactly those, so a driver's `if(me.eq.0)` printing and its nproc banner behave as under mpirun.
subroutine mpi_init(info);            integer :: info; info=0; end subroutine
subroutine mpi_finalize(info);        integer :: info; info=0; end subroutine
subroutine mpi_comm_size(c,n,info)
 integer :: c,n,info,stat
 character(len=32) :: env
 n=1; info=0
 call get_environment_variable('TTX_WORLD_SIZE',env,status=stat)
 if(stat.eq.0)read(env,*)n
end subroutine
subroutine mpi_comm_rank(c,r,info)
 integer :: c,r,info,stat
 character(len=32) :: env
 r=0; info=0
 call get_environment_variable('TTX_WORLD_RANK',env,status=stat)
 if(stat.eq.0)read(env,*)r
end subroutine
