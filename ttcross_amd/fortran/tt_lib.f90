! tt_lib.f90 -- drop-in subset of the reference's ptype_lib / tt_lib (lib/ptype.f90:4-6, lib/tt.f90:16-37,
! 879-892, 908-916, 1228-1245, 1348-1360): the dtt container the drivers and dtt_dmrgg exchange.
! Same component names; one extra hidden component carries the device-engine handle.
module ptype_lib
 implicit none
 type,public :: pointd3
  double precision,dimension(:,:,:),pointer,contiguous :: p=>null()
 end type
 type,public :: pointz3
  double complex,dimension(:,:,:),pointer,contiguous :: p=>null()
 end type
end module

module tt_lib
 use iso_c_binding
 use ptype_lib
 implicit none
 integer,parameter :: tt_size=2048
 type,public :: dtt
  integer :: l=1
  integer :: m=0
  integer :: n(tt_size)=0
  integer :: q(tt_size)=0
  integer :: s(tt_size)=0
  integer :: t=0
  integer :: r(0:tt_size)=0
  type(pointd3) :: u(tt_size)
  type(c_ptr) :: ttx=c_null_ptr      ! engine handle after dtt_dmrgg (not in the reference)
 end type
 ! complex train (lib/tt.f90:38-52): host cores as in the reference; a ztt made from a dtt (tt_z = tt) also BORROWS
 ! the device train of that dtt -- ztt_quad works on it -- and never releases it
 type,public :: ztt
  integer :: l=1
  integer :: m=0
  integer :: n(tt_size)=0
  integer :: q(tt_size)=0
  integer :: s(tt_size)=0
  integer :: t=0
  integer :: r(0:tt_size)=0
  type(pointz3) :: u(tt_size)
  type(c_ptr) :: ttx=c_null_ptr
 end type
 interface alloc;   module procedure dtt_alloc,ztt_alloc;     end interface
 interface dealloc; module procedure dtt_dealloc,ztt_dealloc; end interface
 interface assignment (=); module procedure ztt_dtt_assign; end interface
 interface ones;    module procedure dtt_ones;    end interface
 interface erank;   module procedure dtt_rank;    end interface
 ! utilities that run on the tensor train resident on the device after dtt_dmrgg (lib/tt.f90:54-124 generics)
 interface ort;         module procedure dtt_ort;  end interface
 interface svd;         module procedure dtt_svd;  end interface
 interface norm;        module procedure dtt_norm; end interface
 interface dot_product; module procedure dtt_dot;  end interface
 interface tijk;        module procedure dtt_ijk;  end interface
contains
 subroutine ztt_alloc(arg)
  type(ztt),intent(inout) :: arg
  integer :: k,ierr
  if(arg%m < arg%l) return
  do k = arg%l, arg%m
   if(associated(arg%u(k)%p)) deallocate(arg%u(k)%p)
   allocate(arg%u(k)%p(arg%r(k-1), arg%n(k), arg%r(k)), stat=ierr)
   if(ierr /= 0) then
    write(*,*) 'TT allocate fail: no memory'; stop
   end if
  end do
 end subroutine
 subroutine ztt_dealloc(arg)
  type(ztt),intent(inout) :: arg
  integer :: k
  do k = 1, tt_size
   if(associated(arg%u(k)%p)) then
    deallocate(arg%u(k)%p); nullify(arg%u(k)%p)
   end if
  end do
  arg%ttx = c_null_ptr               ! borrowed, see the type
 end subroutine
 subroutine ztt_dtt_assign(z,d)
  ! lib/tt.f90:1033-1045: complex copy of a real train (plus the borrowed device handle)
  type(ztt),intent(inout) :: z
  type(dtt),intent(in) :: d
  integer :: k
  if(d%l > d%m) then
   write(*,*) 'ztt_dtt_assign: l,m: ',d%l,d%m; return
  end if
  z%l = d%l; z%m = d%m; z%n = d%n; z%r = d%r
  call ztt_alloc(z)
  do k = d%l, d%m
   z%u(k)%p = dcmplx(d%u(k)%p, 0.d0)
  end do
  z%ttx = d%ttx
 end subroutine
 subroutine dtt_alloc(arg)
  ! (re)allocate every core k = l..m with the shape (r(k-1), n(k), r(k))
  type(dtt),intent(inout) :: arg
  integer :: k,ierr
  if(arg%m < arg%l) return
  if(arg%l < 1) then
   write(*,*) 'dtt_alloc: %l should be > 0'; stop
  end if
  if(arg%m > tt_size) then
   write(*,*) 'dtt_alloc: %m exceeds tt_size, change parameter and recompile!'; stop
  end if
  cores: do k = arg%l, arg%m
   if(associated(arg%u(k)%p)) deallocate(arg%u(k)%p)
   allocate(arg%u(k)%p(arg%r(k-1), arg%n(k), arg%r(k)), stat=ierr)
   if(ierr /= 0) then
    write(*,*) 'TT allocate fail: no memory'; stop
   end if
  end do cores
 end subroutine

 subroutine dtt_dealloc(arg)
  ! free the host cores and release the device engine that holds the resident train
  use ttx_c, only: ttx_destroy
  type(dtt),intent(inout) :: arg
  integer :: k
  do k = 1, tt_size
   if(associated(arg%u(k)%p)) then
    deallocate(arg%u(k)%p); nullify(arg%u(k)%p)
   end if
  end do
  if(c_associated(arg%ttx)) then
   call ttx_destroy(arg%ttx)
   arg%ttx = c_null_ptr
  end if
 end subroutine

 subroutine dtt_ones(arg)
  ! rank-one train of ones
  type(dtt),intent(inout) :: arg
  integer :: k
  if(arg%m < arg%l) return
  arg%r(arg%l-1:arg%m) = 1
  call dtt_alloc(arg)
  do k = arg%l, arg%m
   arg%u(k)%p(:,:,:) = 1.d0
  end do
 end subroutine

 subroutine dtt_pull(arg)
  ! refresh arg%r and arg%u from the device-resident tensor train
  use ttx_c
  type(dtt),intent(inout) :: arg
  integer(c_int32_t) :: rk(0:tt_size)
  integer :: k
  call ttx_check(ttx_get_ranks(arg%ttx,rk),'tt_lib')
  arg%r(0:arg%m)=rk(0:arg%m)
  call dtt_alloc(arg)
  do k=1,arg%m; call ttx_check(ttx_get_core(arg%ttx,int(k,c_int),arg%u(k)%p),'tt_lib'); end do
 end subroutine
 subroutine dtt_resident(arg,who)
  type(dtt),intent(in) :: arg
  character(len=*),intent(in) :: who
  if(.not.c_associated(arg%ttx))then;write(*,*)who,': tensor train is not resident on the device (call dtt_dmrgg first)';stop;endif
 end subroutine
 subroutine dtt_ort(arg)
  use ttx_c
  type(dtt),intent(inout),target :: arg
  call dtt_resident(arg,'dtt_ort'); call ttx_check(ttx_ort(arg%ttx),'dtt_ort'); call dtt_pull(arg)
 end subroutine
 subroutine dtt_svd(arg,tol,rmax)
  use ttx_c
  type(dtt),intent(inout),target :: arg
  double precision,intent(in) :: tol
  integer,intent(in),optional :: rmax
  integer(c_int32_t) :: rm
  rm=0; if(present(rmax))rm=rmax
  call dtt_resident(arg,'dtt_svd'); call ttx_check(ttx_svd(arg%ttx,tol,rm),'dtt_svd'); call dtt_pull(arg)
 end subroutine
 double precision function dtt_norm(arg,tol) result(nrm)
  use ttx_c
  type(dtt),intent(in) :: arg
  double precision,intent(in),optional :: tol
  real(c_double) :: t
  t=-1.d0; if(present(tol))t=tol
  call dtt_resident(arg,'dtt_norm'); call ttx_check(ttx_norm(arg%ttx,t,nrm),'dtt_norm')
 end function
 double precision function dtt_dot(x,y) result(dot)
  use ttx_c
  type(dtt),intent(in) :: x,y
  call dtt_resident(x,'dtt_dot'); call dtt_resident(y,'dtt_dot'); call ttx_check(ttx_dot(x%ttx,y%ttx,dot),'dtt_dot')
 end function
 double precision function dtt_ijk(arg,ind) result(a)
  use ttx_c
  type(dtt),intent(in) :: arg
  integer,intent(in) :: ind(:)
  integer(c_int32_t) :: ix(arg%m)
  ix=ind(1:arg%m)
  call dtt_resident(arg,'dtt_ijk'); call ttx_check(ttx_ijk(arg%ttx,ix,a),'dtt_ijk')
 end function
 double precision function dtt_rank(arg) result(er)
  ! effective rank: the r for which a train with all inner ranks r stores as many numbers as arg does,
  ! i.e. the positive root of  a r^2 + b r - S = 0,  S = sum r(k-1) n(k) r(k)
  type(dtt),intent(in) :: arg
  integer :: k,first,last,ncores,inner,edge
  double precision :: stored
  first = arg%l; last = arg%m; ncores = last-first+1
  er = -1.d0
  if(ncores <= 0) return
  er = 0.d0
  if(ncores == 1) return
  stored = 0.d0
  do k = first, last
   stored = stored + arg%r(k-1)*arg%n(k)*arg%r(k)
  end do
  er = stored
  if(stored == 0.d0) return
  edge = arg%r(first-1)*arg%n(first) + arg%n(last)*arg%r(last)
  if(ncores == 2) then
   er = stored/edge
   return
  end if
  inner = sum(arg%n(first+1:last-1))
  er = (dsqrt(edge*edge + 4.d0*inner*stored) - edge)/(2.d0*inner)
 end function
end module
