! dmrgg_lib.f90 -- drop-in for the reference's dmrgg_lib (lib/dmrgg.f90): same module name, same public
! procedures dtt_dmrgg / dtt_quad with the same arguments (lib/dmrgg.f90:11-26, :1261), implemented by the
! MI355X engine through the C-ABI of libttx.so.  A driver written for the reference compiles unchanged:
!     call dtt_dmrgg(tt,dfunc_ising_discr,par,maxrank=r,accuracy=acc,pivoting=piv,neval=neval,quad=qq,tru=tru)
! The user callback `fun` cannot run on the GPU.  It is identified by probing: `fun` is evaluated on the host at
! a few multi-indices and compared with the engine's built-in integrands (Ising C/D/E, stdnorm, mvn).
module dmrgg_lib
 use iso_c_binding
 use tt_lib
 use default_lib
 use time_lib
 use ttx_c
 implicit none
contains
 subroutine dtt_dmrgg(arg,fun,par,accuracy,maxrank,mybonds,pivoting,neval,quad,tru)
  type(dtt),intent(inout),target :: arg
  double precision,external :: fun
  double precision,intent(in),optional,target :: par(*)
  double precision,intent(in),optional :: accuracy
  integer,intent(in),optional :: maxrank
  integer,intent(in),optional :: mybonds(0:)
  integer,intent(in),optional :: pivoting
  integer(kind=8),intent(out),optional :: neval
  type(dtt),intent(in),optional :: quad
  double precision,intent(in),optional :: tru
  character(len=*),parameter :: subnam='dtt_dmrgg'
  type(ttx_config) :: cfg
  integer(c_int32_t),allocatable,target :: nn(:),mb(:),rk(:)
  real(c_double),allocatable,target :: qw(:),aux(:),pcopy(:)
  integer :: l,m,k,off,fid,npar,ngroups,stat
  character(len=32) :: env
  if(arg%l.gt.arg%m)then;write(*,*)subnam,': l,m: ',arg%l,arg%m;stop;endif
  if(arg%l.ne.1)then;write(*,*)subnam,': only l=1 is supported (as in every driver)';stop;endif
  if(.not.present(maxrank))then;write(*,*)subnam,': maxrank is required by the device engine';stop;endif
  if(.not.present(par))then;write(*,*)subnam,': par is required for the built-in integrands';stop;endif
  l=arg%l; m=arg%m
  call identify(fun,m,arg%n,par,fid,npar,aux)
  allocate(nn(m),pcopy(npar)); nn=arg%n(1:m); pcopy=par(1:npar)
  cfg%d=m; cfg%n=c_loc(nn); cfg%fun_id=fid; cfg%par=c_loc(pcopy); cfg%npar=npar
  cfg%aux=c_null_ptr; cfg%naux=0
  if(allocated(aux))then; cfg%aux=c_loc(aux); cfg%naux=size(aux); endif
  cfg%quadw=c_null_ptr
  if(present(quad))then
   allocate(qw(sum(arg%n(1:m)))); off=0
   do k=1,m; qw(off+1:off+arg%n(k))=quad%u(k)%p(1,1:arg%n(k),1); off=off+arg%n(k); end do
   cfg%quadw=c_loc(qw)
  end if
  cfg%accuracy=-1.d0; if(present(accuracy))cfg%accuracy=accuracy
  cfg%maxrank=maxrank
  cfg%pivoting=default(3,pivoting)
  cfg%tru=0.d0; cfg%has_tru=0; if(present(tru))then; cfg%tru=tru; cfg%has_tru=1; endif
  ! bond groups: mybonds(0:nproc) as in the reference, else TTX_NGROUPS groups by share()
  cfg%mybonds=c_null_ptr; ngroups=1
  call get_environment_variable('TTX_NGROUPS',env,status=stat)
  if(stat.eq.0)read(env,*)ngroups
  if(present(mybonds))then
   ngroups=ubound(mybonds,1); allocate(mb(0:ngroups)); mb=mybonds(0:ngroups); cfg%mybonds=c_loc(mb)
  end if
  cfg%nproc=ngroups
  cfg%device=0; cfg%world_rank=0; cfg%world_size=1; cfg%verbose=1; cfg%use_graph=0
  if(c_associated(arg%ttx))then; call ttx_destroy(arg%ttx); arg%ttx=c_null_ptr; endif
  call ttx_check(ttx_create(arg%ttx,cfg),subnam)
  call ttx_check(ttx_run(arg%ttx),subnam)
  ! results back into the caller's container: ranks and finalised cores (ownership as in the reference)
  allocate(rk(0:m))
  call ttx_check(ttx_get_ranks(arg%ttx,rk),subnam)
  arg%r(0:m)=rk(0:m)
  call alloc(arg)
  do k=1,m
   call ttx_check(ttx_get_core(arg%ttx,int(k,c_int),arg%u(k)%p),subnam)
  end do
  if(present(neval))neval=ttx_neval(arg%ttx)
 end subroutine

 double precision function dtt_quad(arg,quad,mybonds) result(val)
  ! lib/dmrgg.f90:1261: rank-1 quadrature of the TT held by the engine (quad absent: sum over all modes)
  type(dtt),intent(in),target :: arg
  type(dtt),intent(in),optional :: quad
  integer,intent(in),optional,target :: mybonds(0:)
  real(c_double),allocatable,target :: qw(:)
  real(c_double) :: v
  integer :: k,off
  if(.not.c_associated(arg%ttx))then;write(*,*)'dtt_quad: tensor train is not resident on the device (call dtt_dmrgg first)';stop;endif
  if(present(quad))then
   allocate(qw(sum(arg%n(1:arg%m)))); off=0
   do k=1,arg%m; qw(off+1:off+arg%n(k))=quad%u(k)%p(1,1:arg%n(k),1); off=off+arg%n(k); end do
   call ttx_check(ttx_quad(arg%ttx,c_loc(qw),v),'dtt_quad')
  else
   call ttx_check(ttx_quad(arg%ttx,c_null_ptr,v),'dtt_quad')
  end if
  val=v
 end function

 double complex function ztt_quad(arg,quad,mybonds) result(val)
  ! lib/dmrgg.f90:1418: quadrature of the (real) device train behind arg with the COMPLEX rank-1 weights in quad
  type(ztt),intent(in),target :: arg
  type(ztt),intent(in),optional :: quad
  integer,intent(in),optional,target :: mybonds(0:)
  real(c_double),allocatable :: w(:)
  real(c_double) :: o(2)
  integer :: k,j,off
  if(.not.c_associated(arg%ttx))then;write(*,*)'ztt_quad: the train was not made from a dtt that is resident on the device';stop;endif
  allocate(w(2*sum(arg%n(1:arg%m)))); off=0
  do k=1,arg%m
   do j=1,arg%n(k)
    if(present(quad))then
     w(off+2*j-1)=dble(quad%u(k)%p(1,j,1)); w(off+2*j)=dimag(quad%u(k)%p(1,j,1))
    else
     w(off+2*j-1)=1.d0; w(off+2*j)=0.d0
    end if
   end do
   off=off+2*arg%n(k)
  end do
  call ttx_check(ttx_zquad(arg%ttx,1_c_int32_t,w,o),'ztt_quad')
  val=dcmplx(o(1),o(2))
 end function

 subroutine dtt_accchk(nlot,arg,einf,efro,ainf,afro,fun,par,pivot)
  ! lib/dmrgg.f90:1081: random-sample error of the TT held by the engine against the integrand it was built from
  integer,intent(in) :: nlot
  type(dtt),intent(in) :: arg
  double precision,intent(out) :: einf,efro,ainf,afro
  double precision,external :: fun
  double precision,intent(inout),optional :: par(*)
  integer,intent(out),optional :: pivot(tt_size)
  integer(c_int32_t) :: pv(tt_size)
  if(.not.c_associated(arg%ttx))then;write(*,*)'dtt_accchk: tensor train is not resident on the device (call dtt_dmrgg first)';stop;endif
  call ttx_check(ttx_accchk(arg%ttx,int(nlot,c_int32_t),einf,efro,ainf,afro,pv),'dtt_accchk')
  if(present(pivot))pivot(1:arg%m)=pv(1:arg%m)
 end subroutine

 subroutine identify(fun,m,n,par,fid,npar,aux)
  ! which built-in integrand is `fun`?  compare values at a few probe indices
  use mvn_pdf_mod
  double precision,external :: fun
  integer,intent(in) :: m,n(*)
  double precision,intent(in) :: par(*)
  integer,intent(out) :: fid,npar
  real(c_double),allocatable,intent(out) :: aux(:)
  integer :: ind(m),t,i,jj,id
  double precision :: f,g,x(m),v,w,vk,wk,a,uij
  logical :: ok(3)
  ok=.true.
  do t=1,3
   do i=1,m; ind(i)=mod(7*t+3*i+i*i*t,n(i))+1; end do
   f=fun(m,ind,n,par)
   ! Ising (test_crs_ising.f90:176-218): needs par(2n+1) in {1,2,3}
   id=0; if(par(2*n(1)+1).ge.1.d0.and.par(2*n(1)+1).le.3.d0)id=int(par(2*n(1)+1))
   if(id.ge.1)then
    a=1.d0
    if(id.ge.2)then
     do i=0,m; uij=1.d0
      do jj=i+1,m; uij=uij*par(ind(jj)); a=a*((uij-1.d0)/(uij+1.d0))**2; end do
     end do
    end if
    v=1.d0;w=1.d0;vk=1.d0;wk=1.d0
    do i=1,m; vk=vk*par(ind(m-i+1)); wk=wk*par(ind(i)); v=v+vk; w=w+wk; end do
    select case(id); case(1);g=2/(v*w); case(2);g=2*a/(v*w); case default;g=2*a; end select
    do i=1,m; g=g*par(n(1)+ind(i)); end do
    if(abs(f-g).gt.1d-12*abs(g))ok(1)=.false.
   else
    ok(1)=.false.
   end if
   do i=1,m; x(i)=par(ind(i)); end do
   g=exp(-sum(x**2)); if(abs(f-g).gt.1d-12*abs(g)+tiny(1.d0))ok(2)=.false.
   if(allocated(mvn_data%mu).and.mvn_data%n.eq.m)then
    g=mvn_pdf(x); if(abs(f-g).gt.1d-12*abs(g)+tiny(1.d0))ok(3)=.false.
   else
    ok(3)=.false.
   end if
  end do
  if(ok(1))then
   fid=TTX_FUN_ISING; npar=2*n(1)+1
  else if(ok(3))then
   fid=TTX_FUN_MVN; npar=2*n(1)
   allocate(aux(m+m*m+1)); aux(1:m)=mvn_data%mu; aux(m+1:m+m*m)=reshape(mvn_data%inv_cov,[m*m]); aux(m+m*m+1)=mvn_data%det_cov
  else if(ok(2))then
   fid=TTX_FUN_STDNORM; npar=2*n(1)
  else
   write(*,*)'dtt_dmrgg: fun is not one of the integrands built into the device engine (Ising C/D/E, stdnorm, mvn)'
   stop
  end if
 end subroutine
end module
