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
 interface assignment (=); module procedure dtt_assign,ztt_assign,ztt_dtt_assign; end interface
 interface ones;    module procedure dtt_ones;    end interface
 interface zeros;   module procedure dtt_zeros;   end interface
 interface erank;   module procedure dtt_rank;    end interface
 ! host-side generics of lib/tt.f90:54-124 (they work on the host cores arg%u(k)%p, as in the reference)
 interface copy;    module procedure dtt_copy;    end interface
 interface ready;   module procedure dtt_ready;   end interface
 interface numel;   module procedure dtt_numel;   end interface
 interface memory;  module procedure dtt_mem;     end interface
 interface say;     module procedure dtt_say;     end interface
 interface sumall;  module procedure dtt_sumall;  end interface
 interface value;   module procedure dtt_value,dtt_value0; end interface
 interface elem;    module procedure dtt_elem;    end interface
 interface lognrm;  module procedure dtt_lognrm;  end interface
 interface operator (+); module procedure dtt_plus_dtt,dtt_plus_d; end interface
 interface operator (*); module procedure dtt_mul_dt; end interface
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
  call dtt_release(arg)
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
 subroutine dtt_stage(arg,h,temp,who)
  ! device handle for a utility call: the resident train of arg (after dtt_dmrgg / read / ort / svd), else a temporary
  ! upload of the host cores -- trains built on the host (ones, =, +, *) carry no handle and are staged per call,
  ! so editing arg%u(k)%p between calls is always seen
  use ttx_c
  type(dtt),intent(in) :: arg
  type(c_ptr),intent(out) :: h
  logical,intent(out) :: temp
  character(len=*),intent(in) :: who
  integer(c_int32_t) :: nn(tt_size),rr(0:tt_size)
  double precision,allocatable :: x(:)
  integer :: k,sz,pos
  temp=.false.; h=arg%ttx
  if(c_associated(h))return
  if(arg%l.ne.1 .or. arg%m.lt.2)then;write(*,*)who,': the device engine holds trains with l=1, m>=2; got l,m: ',arg%l,arg%m;stop;endif
  if(.not.dtt_ready(arg))then;write(*,*)who,': tensor train is not allocated';stop;endif
  sz=0; do k=1,arg%m; sz=sz+arg%r(k-1)*arg%n(k)*arg%r(k); end do
  allocate(x(sz)); pos=0
  do k=1,arg%m
   x(pos+1:pos+size(arg%u(k)%p))=reshape(arg%u(k)%p,[size(arg%u(k)%p)]); pos=pos+size(arg%u(k)%p)
  end do
  nn(1:arg%m)=arg%n(1:arg%m); rr(0:arg%m)=arg%r(0:arg%m)
  call ttx_check(ttx_from_tt(h,int(arg%m,c_int32_t),nn,rr,x,0_c_int32_t),who)
  temp=.true.
 end subroutine
 subroutine dtt_ort(arg)
  use ttx_c
  type(dtt),intent(inout),target :: arg
  type(c_ptr) :: h
  logical :: temp
  call dtt_stage(arg,h,temp,'dtt_ort'); call ttx_check(ttx_ort(h),'dtt_ort')
  arg%ttx=h; call dtt_pull(arg)
  if(temp)then; call ttx_destroy(h); arg%ttx=c_null_ptr; endif
 end subroutine
 subroutine dtt_svd(arg,tol,rmax)
  use ttx_c
  type(dtt),intent(inout),target :: arg
  double precision,intent(in) :: tol
  integer,intent(in),optional :: rmax
  integer(c_int32_t) :: rm
  type(c_ptr) :: h
  logical :: temp
  rm=0; if(present(rmax))rm=rmax
  call dtt_stage(arg,h,temp,'dtt_svd'); call ttx_check(ttx_svd(h,tol,rm),'dtt_svd')
  arg%ttx=h; call dtt_pull(arg)
  if(temp)then; call ttx_destroy(h); arg%ttx=c_null_ptr; endif
 end subroutine
 double precision function dtt_norm(arg,tol) result(nrm)
  use ttx_c
  type(dtt),intent(in) :: arg
  double precision,intent(in),optional :: tol
  real(c_double) :: t
  type(c_ptr) :: h
  logical :: temp
  t=-1.d0; if(present(tol))t=tol
  call dtt_stage(arg,h,temp,'dtt_norm'); call ttx_check(ttx_norm(h,t,nrm),'dtt_norm')
  if(temp)call ttx_destroy(h)
 end function
 double precision function dtt_lognrm(arg,tol) result(nrm)
  ! lib/tt.f90:1114: log10 of the Frobenius norm
  type(dtt),intent(in) :: arg
  double precision,intent(in),optional :: tol
  nrm=log10(dtt_norm(arg,tol))
 end function
 double precision function dtt_dot(x,y) result(dot)
  use ttx_c
  type(dtt),intent(in) :: x,y
  type(c_ptr) :: hx,hy
  logical :: tx,ty
  call dtt_stage(x,hx,tx,'dtt_dot'); call dtt_stage(y,hy,ty,'dtt_dot'); call ttx_check(ttx_dot(hx,hy,dot),'dtt_dot')
  if(tx)call ttx_destroy(hx)
  if(ty)call ttx_destroy(hy)
 end function
 double precision function dtt_ijk(arg,ind) result(a)
  ! lib/tt.f90:630-652: one element; a resident train is asked on the device, a host train is contracted here
  use ttx_c
  type(dtt),intent(in) :: arg
  integer,intent(in) :: ind(:)
  integer(c_int32_t) :: ix(arg%m)
  double precision :: b(1)
  if(c_associated(arg%ttx))then
   ix=ind(1:arg%m)
   call ttx_check(ttx_ijk(arg%ttx,ix,a),'dtt_ijk')
  else
   if(arg%r(arg%l-1).ne.1 .or. arg%r(arg%m).ne.1)then; a=-4.d0; return; endif
   if(any(ind(1:arg%m-arg%l+1).le.0).or.any(ind(1:arg%m-arg%l+1).gt.arg%n(arg%l:arg%m)))then; a=-3.d0; return; endif
   call dtt_elem(arg,ind,b); a=b(1)
  end if
 end function

 ! ---- host-side generics (lib/tt.f90): plain Fortran on the host cores ----------------------------------
 logical function dtt_ready(arg) result(l)
  ! lib/tt.f90:1306: dimensions set and every core allocated with the right shape
  type(dtt),intent(in) :: arg
  integer :: k
  l=.false.
  if(arg%l.gt.arg%m)return
  do k=arg%l,arg%m
   if(arg%n(k).le.0 .or. arg%r(k-1).le.0 .or. arg%r(k).le.0)return
   if(.not.associated(arg%u(k)%p))return
   if(any(shape(arg%u(k)%p).ne.[arg%r(k-1),arg%n(k),arg%r(k)]))return
  end do
  l=.true.
 end function
 subroutine dtt_release(arg)
  ! drop the device train of arg (it is about to receive new contents)
  use ttx_c, only: ttx_destroy
  type(dtt),intent(inout) :: arg
  if(c_associated(arg%ttx))then; call ttx_destroy(arg%ttx); arg%ttx=c_null_ptr; endif
 end subroutine
 subroutine dtt_assign(b,a)
  ! lib/tt.f90:1012-1020: b = a is a DEEP copy of the cores.  The device train stays with a: b is a host train (its own
  ! storage, no handle), so dealloc(a) and dealloc(b) each release only what they own
  type(dtt),intent(inout) :: b
  type(dtt),intent(in) :: a
  integer :: k
  call dtt_release(b)
  b%l=a%l; b%m=a%m; b%n(a%l:a%m)=a%n(a%l:a%m); b%r(a%l-1:a%m)=a%r(a%l-1:a%m)
  call dtt_alloc(b)
  do k=a%l,a%m; b%u(k)%p=a%u(k)%p; end do
 end subroutine
 subroutine ztt_assign(b,a)
  ! lib/tt.f90:1021-1032
  type(ztt),intent(inout) :: b
  type(ztt),intent(in) :: a
  integer :: k
  b%l=a%l; b%m=a%m; b%n(a%l:a%m)=a%n(a%l:a%m); b%r(a%l-1:a%m)=a%r(a%l-1:a%m)
  call ztt_alloc(b)
  do k=a%l,a%m; b%u(k)%p=a%u(k)%p; end do
  b%ttx=a%ttx                        ! borrowed from the dtt it was made from, never released by a ztt
 end subroutine
 subroutine dtt_copy(a,b,low)
  ! lib/tt.f90:1047-1058: copy a into b with the first core placed at index low (default b%l)
  type(dtt),intent(in) :: a
  type(dtt),intent(inout) :: b
  integer,intent(in),optional :: low
  integer :: k,ll,mm
  ll=b%l; if(present(low))ll=low
  mm=ll-a%l+a%m
  call dtt_release(b)
  b%l=ll; b%m=mm; b%n(ll:mm)=a%n(a%l:a%m); b%r(ll-1:mm)=a%r(a%l-1:a%m)
  if(.not.all(a%n(a%l:a%m)>0))return
  if(.not.all(a%r(a%l-1:a%m)>0))return
  call dtt_alloc(b)
  do k=a%l,a%m; b%u(ll-a%l+k)%p=a%u(k)%p; end do
 end subroutine
 subroutine dtt_zeros(arg)
  ! lib/tt.f90:1375: rank-one train of zeros
  type(dtt),intent(inout) :: arg
  integer :: k
  if(arg%m < arg%l) return
  call dtt_release(arg)
  arg%r(arg%l-1:arg%m)=1
  call dtt_alloc(arg)
  do k=arg%l,arg%m; arg%u(k)%p=0.d0; end do
 end subroutine
 double precision function dtt_numel(arg) result(s)
  ! lib/tt.f90:817: number of entries of the full tensor
  type(dtt),intent(in) :: arg
  integer :: k
  s=0.d0; if(arg%l.gt.arg%m)return
  s=1.d0; do k=arg%l,arg%m; s=s*arg%n(k); end do
 end function
 integer function dtt_mem(arg) result(sz)
  ! lib/tt.f90:1266: numbers stored in the cores
  type(dtt),intent(in) :: arg
  integer :: k
  sz=0; do k=arg%l,arg%m; sz=sz+arg%r(k-1)*arg%n(k)*arg%r(k); end do
 end function
 subroutine dtt_say(arg)
  ! lib/tt.f90:1200: short description on stdout
  type(dtt),intent(in) :: arg
  write(*,'(a,i2,a,i4,a,f6.2,a,i12)') 'dtt[',arg%l,':',arg%m,']: rank ',dtt_rank(arg),' memory ',dtt_mem(arg)
  write(*,'(a,1x,64i4)') 'n: ',arg%n(arg%l:min(arg%m,arg%l+63))
  write(*,'(a,64i4)') 'r: ',arg%r(arg%l-1:min(arg%m,arg%l+62))
 end subroutine
 subroutine dtt_elem(arg,ind,a)
  ! lib/tt.f90:678-700: the r(l-1) x r(m) block of the train at a multi-index, left to right
  type(dtt),intent(in) :: arg
  integer,intent(in) :: ind(:)
  double precision,intent(out) :: a(*)
  double precision,allocatable :: x(:,:),z(:,:)
  integer :: k,l,m
  l=arg%l; m=arg%m
  if(any(ind(1:m-l+1).le.0).or.any(ind(1:m-l+1).gt.arg%n(l:m)))then;write(*,*)'dtt_elem: wrong index: ',ind;stop;endif
  allocate(x(arg%r(l-1),arg%r(l))); x=arg%u(l)%p(:,ind(1),:)
  do k=l+1,m
   allocate(z(arg%r(l-1),arg%r(k))); z=matmul(x,arg%u(k)%p(:,ind(k-l+1),:))
   call move_alloc(z,x)
  end do
  a(1:arg%r(l-1)*arg%r(m))=reshape(x,[arg%r(l-1)*arg%r(m)])
 end subroutine
 double precision function dtt_value(arg,x) result(val)
  ! lib/tt.f90:702-731: the train read as a function on [0,1]^dd, each coordinate spread over (m-l+1)/dd modes
  ! (most significant digit in the LAST mode of its block)
  type(dtt),intent(in) :: arg
  double precision,intent(in) :: x(:)
  integer :: dd,per,id,j,pos,i,ind(tt_size)
  double precision :: xx
  ind=0; val=0.d0; dd=size(x)
  if(arg%l.gt.arg%m)return
  per=(arg%m-arg%l+1)/dd
  do id=1,dd
   xx=x(id)
   if(xx.lt.0.d0)return
   if(xx.gt.1.d0)xx=xx-int(xx)
   do j=1,per
    pos=arg%l+(id-1)*per+per-j
    i=int(arg%n(pos)*xx); if(i.eq.arg%n(pos))i=arg%n(pos)-1
    ind(pos-arg%l+1)=i+1
    xx=xx*arg%n(pos)-i
   end do
  end do
  val=dtt_ijk(arg,ind(1:arg%m-arg%l+1))
 end function
 double precision function dtt_value0(arg,x) result(val)
  type(dtt),intent(in) :: arg
  double precision,intent(in) :: x
  val=dtt_value(arg,[x])
 end function
 double precision function dtt_sumall(arg) result(val)
  ! lib/tt.f90:770-790: sum of all entries = product of the mode-summed cores, left to right
  type(dtt),intent(in) :: arg
  double precision,allocatable :: x(:,:),z(:,:)
  integer :: k,l,m
  l=arg%l; m=arg%m; val=0.d0
  if(arg%r(l-1).gt.1 .or. arg%r(m).gt.1)then; val=-1.d0; return; endif
  allocate(x(arg%r(l-1),arg%r(l))); x=sum(arg%u(l)%p,dim=2)
  do k=l+1,m
   allocate(z(arg%r(l-1),arg%r(k))); z=matmul(x,sum(arg%u(k)%p,dim=2))
   call move_alloc(z,x)
  end do
  val=x(1,1)
 end function
 function dtt_plus_dtt(a,b) result(c)
  ! lib/tt.f90:928-946: c = a + b, ranks add, cores become block diagonal (first core: side by side, last: stacked)
  type(dtt),intent(in) :: a,b
  type(dtt) :: c
  integer :: k,l,m,ra0,ra1
  if(.not.(dtt_ready(a).and.dtt_ready(b)))return
  l=a%l; m=a%m
  c%l=l; c%m=m; c%n=a%n; c%r=0; c%r(l-1)=a%r(l-1); c%r(m)=a%r(m); c%r(l:m-1)=a%r(l:m-1)+b%r(l:m-1)
  call dtt_alloc(c)
  do k=l,m
   c%u(k)%p=0.d0
   ra0=a%r(k-1); ra1=a%r(k)
   if(k.eq.l)then
    c%u(k)%p(:,:,1:ra1)=a%u(k)%p; if(m.gt.l)c%u(k)%p(:,:,ra1+1:)=b%u(k)%p
   else if(k.eq.m)then
    c%u(k)%p(1:ra0,:,:)=a%u(k)%p; c%u(k)%p(ra0+1:,:,:)=b%u(k)%p
   else
    c%u(k)%p(1:ra0,:,1:ra1)=a%u(k)%p; c%u(k)%p(ra0+1:,:,ra1+1:)=b%u(k)%p
   end if
  end do
 end function
 function dtt_plus_d(a,b) result(c)
  ! lib/tt.f90:966-986: c = a + b*ones
  type(dtt),intent(in) :: a
  double precision,intent(in) :: b
  type(dtt) :: c,e
  integer :: k
  if(.not.dtt_ready(a))return
  e%l=a%l; e%m=a%m; e%n=a%n; call dtt_ones(e); e%u(e%l)%p=b
  c=dtt_plus_dtt(a,e)
  do k=e%l,e%m; deallocate(e%u(k)%p); end do
 end function
 function dtt_mul_dt(a,b) result(c)
  ! lib/tt.f90:989-998: c = a*b, the scalar goes into the first core
  double precision,intent(in) :: a
  type(dtt),intent(in) :: b
  type(dtt) :: c
  integer :: k
  c%l=b%l; c%m=b%m; c%n=b%n; c%r=b%r; call dtt_alloc(c)
  do k=b%l,b%m; c%u(k)%p=b%u(k)%p; end do
  c%u(b%l)%p=a*c%u(b%l)%p
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
