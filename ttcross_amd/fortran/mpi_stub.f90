! one-rank MPI stand-in (see mpif.h)
subroutine mpi_init(info);            integer :: info; info=0; end subroutine
subroutine mpi_finalize(info);        integer :: info; info=0; end subroutine
subroutine mpi_comm_size(c,n,info);   integer :: c,n,info; n=1; info=0; end subroutine
subroutine mpi_comm_rank(c,r,info);   integer :: c,r,info; r=0; info=0; end subroutine
