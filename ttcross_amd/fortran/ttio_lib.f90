! ttio_lib.f90 -- drop-in for the reference's ttio_lib (lib/ttio.f90:19-24, 29-108, 196-297): `call write(tt,fnam,info)`
! and `call read(tt,fnam,info)` of the raw 'TT      ' stream file.  The file is produced / parsed by the engine
! (ttx_write / ttx_read, include/ttx.h); a train that was read is resident on the device, so ort / svd / norm /
! dot_product / tijk / dtt_quad work on it directly.
module ttio_lib
 use iso_c_binding
 use tt_lib
 implicit none
 interface read;  module procedure dtt_read;  end interface
 interface write; module procedure dtt_write; end interface
contains
 function cstr(s) result(c)
  character(len=*),intent(in) :: s
  character(kind=c_char) :: c(len_trim(s)+1)
  integer :: i
  do i=1,len_trim(s); c(i)=s(i:i); end do
  c(len_trim(s)+1)=c_null_char
 end function

 subroutine dtt_write(arg,fnam,info)
  use ttx_c
  type(dtt),intent(in) :: arg
  character(len=*),intent(in) :: fnam
  integer,intent(out),optional :: info
  type(c_ptr) :: h
  integer(c_int32_t) :: nn(tt_size),rr(0:tt_size)
  double precision,allocatable :: x(:)
  integer :: k,sz,pos,rc
  if(present(info))info=-11
  if(c_associated(arg%ttx))then
   rc=ttx_write(arg%ttx,cstr(fnam))
  else
   ! host-only train (e.g. after ones()): stage it on the device, write, release
   if(arg%l.ne.1 .or. arg%m.lt.2)then;write(*,*)'dtt_write: only trains with l=1, m>=2 are supported: ',arg%l,arg%m;return;endif
   sz=0; do k=1,arg%m; sz=sz+arg%r(k-1)*arg%n(k)*arg%r(k); end do
   if(sz.le.0)then;write(*,*)'dtt_write: tt structure has invalid size: ',sz;return;endif
   allocate(x(sz)); pos=0
   do k=1,arg%m
    x(pos+1:pos+size(arg%u(k)%p))=reshape(arg%u(k)%p,[size(arg%u(k)%p)]); pos=pos+size(arg%u(k)%p)
   end do
   nn(1:arg%m)=arg%n(1:arg%m); rr(0:arg%m)=arg%r(0:arg%m)
   call ttx_check(ttx_from_tt(h,int(arg%m,c_int32_t),nn,rr,x,0_c_int32_t),'dtt_write')
   rc=ttx_write(h,cstr(fnam))
   call ttx_destroy(h)
   deallocate(x)
  end if
  if(rc.ne.0)then
   call ttx_warn('dtt_write'); if(present(info))info=-1
  else
   if(present(info))info=0
  end if
 end subroutine

 subroutine dtt_read(arg,fnam,info)
  use ttx_c
  type(dtt),intent(inout) :: arg
  character(len=*),intent(in) :: fnam
  integer,intent(out),optional :: info
  type(c_ptr) :: h
  integer(c_int32_t) :: d,nn(tt_size)
  integer :: rc
  if(present(info))info=-11
  rc=ttx_read(h,cstr(fnam),0_c_int32_t)
  if(rc.ne.0)then
   call ttx_warn('dtt_read'); if(present(info))info=-1
   return
  end if
  call dealloc(arg)
  call ttx_check(ttx_get_modes(h,d,nn),'dtt_read')
  arg%l=1; arg%m=d; arg%n=0; arg%n(1:d)=nn(1:d); arg%ttx=h
  call dtt_pull(arg)
  if(present(info))info=0
 end subroutine
end module
