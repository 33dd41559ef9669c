! ttx_c.f90 -- ISO_C_BINDING view of include/ttx.h (the C-ABI of libttx.so).
! This is the binding a maintainer of the reference adds to call the MI355X engine from lib/dmrgg.f90.
module ttx_c
 use iso_c_binding
 implicit none
 integer(c_int32_t),parameter :: TTX_FUN_ISING=1, TTX_FUN_STDNORM=2, TTX_FUN_MVN=3, TTX_FUN_HOST=4
 type,bind(C) :: ttx_config
  integer(c_int32_t) :: d
  type(c_ptr) :: n
  integer(c_int32_t) :: fun_id
  type(c_ptr) :: par
  integer(c_int32_t) :: npar
  type(c_ptr) :: aux
  integer(c_int32_t) :: naux
  type(c_ptr) :: quadw
  real(c_double) :: accuracy
  integer(c_int32_t) :: maxrank
  integer(c_int32_t) :: pivoting
  real(c_double) :: tru
  integer(c_int32_t) :: has_tru
  integer(c_int32_t) :: nproc
  type(c_ptr) :: mybonds
  integer(c_int32_t) :: device
  integer(c_int32_t) :: world_rank
  integer(c_int32_t) :: world_size
  integer(c_int32_t) :: verbose
  integer(c_int32_t) :: arith
 end type
 interface
  function ttx_last_error() bind(C,name='ttx_last_error') result(p)
   import; type(c_ptr) :: p
  end function
  function ttx_create(h,cfg) bind(C,name='ttx_create') result(rc)
   import; type(c_ptr),intent(out) :: h; type(ttx_config),intent(in) :: cfg; integer(c_int) :: rc
  end function
  subroutine ttx_destroy(h) bind(C,name='ttx_destroy')
   import; type(c_ptr),value :: h
  end subroutine
  function ttx_set_integrand_host(h,fun,par) bind(C,name='ttx_set_integrand_host') result(rc)   ! the user's `fun`, lib/dmrgg.f90:18
   import; type(c_ptr),value :: h; type(c_funptr),value :: fun; type(c_ptr),value :: par; integer(c_int) :: rc
  end function
  function ttx_getppid() bind(C,name='getppid') result(p)      ! libc: the launcher's pid, shared by the ranks of a job
   import; integer(c_int) :: p
  end function
  function ttx_usleep(us) bind(C,name='usleep') result(rc)
   import; integer(c_int32_t),value :: us; integer(c_int) :: rc
  end function
  function ttx_fun_id(h) bind(C,name='ttx_fun_id') result(id)
   import; type(c_ptr),value :: h; integer(c_int) :: id
  end function
  function ttx_host_calls(h) bind(C,name='ttx_host_calls') result(n)
   import; type(c_ptr),value :: h; integer(c_int64_t) :: n
  end function
  function ttx_comm_unique_id(id) bind(C,name='ttx_comm_unique_id') result(rc)   ! replaces mpi_init, test_crs_ising.f90:31-36
   import; integer(c_int8_t) :: id(128); integer(c_int) :: rc
  end function
  function ttx_comm_init(h,id) bind(C,name='ttx_comm_init') result(rc)
   import; type(c_ptr),value :: h; integer(c_int8_t) :: id(128); integer(c_int) :: rc
  end function
  function ttx_comm_init_shm(h,name) bind(C,name='ttx_comm_init_shm') result(rc)   ! node-local host transport (several ranks on one GPU, no MPI)
   import; type(c_ptr),value :: h; character(kind=c_char) :: name(*); integer(c_int) :: rc
  end function
  function ttx_run(h) bind(C,name='ttx_run') result(rc)
   import; type(c_ptr),value :: h; integer(c_int) :: rc
  end function
  function ttx_neval(h) bind(C,name='ttx_neval') result(n)
   import; type(c_ptr),value :: h; integer(c_int64_t) :: n
  end function
  function ttx_get_ranks(h,r) bind(C,name='ttx_get_ranks') result(rc)
   import; type(c_ptr),value :: h; integer(c_int32_t),intent(out) :: r(*); integer(c_int) :: rc
  end function
  function ttx_core_size(h,k) bind(C,name='ttx_core_size') result(n)
   import; type(c_ptr),value :: h; integer(c_int),value :: k; integer(c_int64_t) :: n
  end function
  function ttx_get_core(h,k,buf) bind(C,name='ttx_get_core') result(rc)
   import; type(c_ptr),value :: h; integer(c_int),value :: k; real(c_double),intent(out) :: buf(*); integer(c_int) :: rc
  end function
  function ttx_quad(h,w,val) bind(C,name='ttx_quad') result(rc)
   import; type(c_ptr),value :: h; type(c_ptr),value :: w; real(c_double),intent(out) :: val; integer(c_int) :: rc
  end function
  function ttx_ort(h) bind(C,name='ttx_ort') result(rc)
   import; type(c_ptr),value :: h; integer(c_int) :: rc
  end function
  function ttx_svd(h,tol,rmax) bind(C,name='ttx_svd') result(rc)
   import; type(c_ptr),value :: h; real(c_double),value :: tol; integer(c_int32_t),value :: rmax; integer(c_int) :: rc
  end function
  function ttx_norm(h,tol,val) bind(C,name='ttx_norm') result(rc)
   import; type(c_ptr),value :: h; real(c_double),value :: tol; real(c_double),intent(out) :: val; integer(c_int) :: rc
  end function
  function ttx_dot(hx,hy,val) bind(C,name='ttx_dot') result(rc)
   import; type(c_ptr),value :: hx,hy; real(c_double),intent(out) :: val; integer(c_int) :: rc
  end function
  function ttx_ijk(h,ind,val) bind(C,name='ttx_ijk') result(rc)
   import; type(c_ptr),value :: h; integer(c_int32_t),intent(in) :: ind(*); real(c_double),intent(out) :: val; integer(c_int) :: rc
  end function
  function ttx_accchk(h,nlot,einf,efro,ainf,afro,pivot) bind(C,name='ttx_accchk') result(rc)
   import; type(c_ptr),value :: h; integer(c_int32_t),value :: nlot; real(c_double),intent(out) :: einf,efro,ainf,afro
   integer(c_int32_t),intent(out) :: pivot(*); integer(c_int) :: rc
  end function
  function ttx_zquad(h,nf,w,out) bind(C,name='ttx_zquad') result(rc)
   import; type(c_ptr),value :: h; integer(c_int32_t),value :: nf; real(c_double) :: w(*),out(*); integer(c_int) :: rc
  end function
  function ttx_from_tt(h,d,n,r,cores,device) bind(C,name='ttx_from_tt') result(rc)
   import; type(c_ptr) :: h; integer(c_int32_t),value :: d,device; integer(c_int32_t) :: n(*),r(*); real(c_double) :: cores(*); integer(c_int) :: rc
  end function
  function ttx_write(h,path) bind(C,name='ttx_write') result(rc)
   import; type(c_ptr),value :: h; character(kind=c_char) :: path(*); integer(c_int) :: rc
  end function
  function ttx_read(h,path,device) bind(C,name='ttx_read') result(rc)
   import; type(c_ptr) :: h; character(kind=c_char) :: path(*); integer(c_int32_t),value :: device; integer(c_int) :: rc
  end function
  function ttx_get_modes(h,d,n) bind(C,name='ttx_get_modes') result(rc)
   import; type(c_ptr),value :: h; integer(c_int32_t) :: d; integer(c_int32_t) :: n(*); integer(c_int) :: rc
  end function
 end interface
contains
 subroutine ttx_warn(who)
  ! print the engine's last error without stopping (the reference's I/O routines report and return, lib/ttio.f90:85-108)
  character(len=*),intent(in) :: who
  character(kind=c_char),pointer :: s(:)
  integer :: i
  call c_f_pointer(ttx_last_error(),s,[512])
  write(*,'(a,a)',advance='no') who,': '
  do i=1,512
   if(s(i).eq.c_null_char)exit
   write(*,'(a)',advance='no') s(i)
  end do
  write(*,*)
 end subroutine
 subroutine ttx_check(rc,who)
  ! the reference reports errors as `write(*,*) ...; stop` (e.g. lib/dmrgg.f90:88-91,105-117)
  integer(c_int),intent(in) :: rc
  character(len=*),intent(in) :: who
  character(kind=c_char),pointer :: s(:)
  integer :: i
  if(rc.eq.0)return
  call c_f_pointer(ttx_last_error(),s,[512])
  write(*,'(a,a)',advance='no') who,': '
  do i=1,512
   if(s(i).eq.c_null_char)exit
   write(*,'(a)',advance='no') s(i)
  end do
  write(*,*)
  stop 1
 end subroutine
end module
