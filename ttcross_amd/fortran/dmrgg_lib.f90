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
 private :: identify,ising_value,ttx_world,ttx_comm_init_file,ttx_comm_init_shm_env
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
  integer :: l,m,k,off,fid,npar,ngroups,stat,rmx,wrank,wsize
  character(len=32) :: env
  if(arg%l.gt.arg%m)then;write(*,*)subnam,': l,m: ',arg%l,arg%m;stop;endif
  if(arg%l.ne.1)then;write(*,*)subnam,': only l=1 is supported (as in every driver)';stop;endif
  l=arg%l; m=arg%m
  ! which integrand?  One of the device integrands when `fun` is recognised (or named by TTX_INTEGRAND), else the
  ! user's `fun` itself, evaluated on the host (the reference's contract, lib/dmrgg.f90:18)
  if(present(par))then
   call identify(fun,m,arg%n,par,fid,npar,aux,pcopy)
  else
   fid=TTX_FUN_HOST; npar=0
  end if
  allocate(nn(m)); nn=arg%n(1:m)
  cfg%d=m; cfg%n=c_loc(nn); cfg%fun_id=fid; cfg%par=c_null_ptr; cfg%npar=npar
  if(npar.gt.0)cfg%par=c_loc(pcopy)
  cfg%aux=c_null_ptr; cfg%naux=0
  if(allocated(aux))then; cfg%aux=c_loc(aux); cfg%naux=size(aux); endif
  cfg%quadw=c_null_ptr
  if(present(quad))then
   allocate(qw(sum(arg%n(1:m)))); off=0
   do k=1,m; qw(off+1:off+arg%n(k))=quad%u(k)%p(1,1:arg%n(k),1); off=off+arg%n(k); end do
   cfg%quadw=c_loc(qw)
  end if
  cfg%accuracy=-1.d0; if(present(accuracy))cfg%accuracy=accuracy
  ! maxrank is optional in the reference (:19-21, :312: without it only the accuracy rule stops the sweeps); the device
  ! engine sizes its storage by it, so an absent maxrank becomes the engine's largest rank (TTX_MAXRANK_DEFAULT, 128)
  rmx=128
  call get_environment_variable('TTX_MAXRANK_DEFAULT',env,status=stat)
  if(stat.eq.0)read(env,*)rmx
  if(present(maxrank))rmx=maxrank
  cfg%maxrank=rmx
  cfg%pivoting=default(3,pivoting)
  cfg%tru=0.d0; cfg%has_tru=0; if(present(tru))then; cfg%tru=tru; cfg%has_tru=1; endif
  ! bond groups: mybonds(0:nproc) as in the reference, else TTX_NGROUPS groups by share()
  cfg%mybonds=c_null_ptr; ngroups=1
  call get_environment_variable('TTX_NGROUPS',env,status=stat)
  if(stat.eq.0)read(env,*)ngroups
  if(present(mybonds))then
   ngroups=ubound(mybonds,1); allocate(mb(0:ngroups)); mb=mybonds(0:ngroups); cfg%mybonds=c_loc(mb)
  end if
  ! one process per GPU (the reference: one MPI rank per bond group): the launcher exports TTX_WORLD_RANK / TTX_WORLD_SIZE
  ! (mpi_stub.f90 reads the same variables for mpi_comm_rank / mpi_comm_size); groups are dealt contiguously to the processes
  call ttx_world(wrank,wsize)
  if(wsize.gt.1 .and. .not.present(mybonds) .and. ngroups.lt.wsize)ngroups=wsize
  cfg%nproc=ngroups
  cfg%device=0; cfg%world_rank=wrank; cfg%world_size=wsize; cfg%verbose=1; cfg%arith=0
  call get_environment_variable('TTX_DEVICE',env,status=stat)
  if(stat.eq.0)read(env,*)cfg%device
  if(c_associated(arg%ttx))then; call ttx_destroy(arg%ttx); arg%ttx=c_null_ptr; endif
  call ttx_check(ttx_create(arg%ttx,cfg),subnam)
  if(fid.eq.TTX_FUN_HOST)then
   if(present(par))then
    call ttx_check(ttx_set_integrand_host(arg%ttx,c_funloc(fun),c_loc(par)),subnam)
   else
    call ttx_check(ttx_set_integrand_host(arg%ttx,c_funloc(fun),c_null_ptr),subnam)
   end if
  end if
  if(wsize.gt.1)then
   ! transport between the processes: RCCL over xGMI (unique id through the file TTX_COMM_FILE), or with TTX_TRANSPORT=shm
   ! the engine's node-local shared-memory transport (several ranks on one GPU, where RCCL cannot run)
   call get_environment_variable('TTX_TRANSPORT',env,status=stat)
   if(stat.eq.0 .and. trim(env).eq.'shm')then
    call ttx_check(ttx_comm_init_shm_env(arg%ttx),subnam)
   else
    call ttx_check(ttx_comm_init_file(arg%ttx),subnam)
   end if
  end if
  call ttx_check(ttx_run(arg%ttx),subnam)
  ! results back into the caller's container: ranks and finalised cores (ownership as in the reference: each process
  ! holds the cores of its own groups; the others stay allocated at the global ranks)
  allocate(rk(0:m))
  call ttx_check(ttx_get_ranks(arg%ttx,rk),subnam)
  arg%r(0:m)=rk(0:m)
  call alloc(arg)
  do k=1,m
   if(ttx_core_size(arg%ttx,int(k,c_int)).gt.0)call ttx_check(ttx_get_core(arg%ttx,int(k,c_int),arg%u(k)%p),subnam)
  end do
  if(.not.present(maxrank))then
   if(maxval(rk(0:m)).ge.rmx)write(*,*)subnam,': rank limit of the device engine reached before the accuracy rule fired: ',rmx
  end if
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
  double precision,intent(inout),optional,target :: par(*)
  integer,intent(out),optional :: pivot(tt_size)
  integer(c_int32_t) :: pv(tt_size)
  if(.not.c_associated(arg%ttx))then;write(*,*)'dtt_accchk: tensor train is not resident on the device (call dtt_dmrgg first)';stop;endif
  ! a host-evaluated integrand is checked against the fun / par given HERE, as in the reference (:1081-1166); the pointers that
  ! dtt_dmrgg registered may belong to a temporary or to storage the caller has released since
  if(ttx_fun_id(arg%ttx).eq.TTX_FUN_HOST)then
   if(present(par))then
    call ttx_check(ttx_set_integrand_host(arg%ttx,c_funloc(fun),c_loc(par)),'dtt_accchk')
   else
    call ttx_check(ttx_set_integrand_host(arg%ttx,c_funloc(fun),c_null_ptr),'dtt_accchk')
   end if
  end if
  call ttx_check(ttx_accchk(arg%ttx,int(nlot,c_int32_t),einf,efro,ainf,afro,pv),'dtt_accchk')
  if(present(pivot))pivot(1:arg%m)=pv(1:arg%m)
 end subroutine

 subroutine identify(fun,m,n,par,fid,npar,aux,pcopy)
  ! Which integrand is `fun`?  TTX_INTEGRAND = ising | stdnorm | mvn | host names it; otherwise (auto) `fun` is compared
  ! with the five device integrands at NPROBE multi-indices spread over the whole index range.  The comparison FAILS
  ! CLOSED: a device integrand is taken only if every probe agrees to 1e-12 RELATIVE, at least three probes are far
  ! above underflow, and exactly one candidate qualifies -- anything else is the user's own function and runs through
  ! the host callback.  The probes read par(1:2n) only (nodes, weights: what every driver of the scope allocates);
  ! the Ising kind C/D/E is inferred from the values, so par(2n+1) is never touched here.
  use mvn_pdf_mod
  double precision,external :: fun
  integer,intent(in) :: m,n(*)
  double precision,intent(in) :: par(*)
  integer,intent(out) :: fid,npar
  real(c_double),allocatable,intent(out) :: aux(:)
  real(c_double),allocatable,target,intent(out) :: pcopy(:)
  integer,parameter :: NPROBE=8
  integer :: ind(m),t,i,c,good(5),hits,stat,pick,n1
  double precision :: f,g(5),x(m)
  logical :: ok(5),have(5)
  character(len=32) :: env
  n1=n(1)
  call get_environment_variable('TTX_INTEGRAND',env,status=stat)
  if(stat.ne.0)env='auto'
  pick=0
  select case(trim(env))
   case('host'); fid=TTX_FUN_HOST; npar=0; return
   case('ising'); pick=int(par(2*n1+1))
    if(pick.lt.1.or.pick.gt.3)then;write(*,*)'dtt_dmrgg: TTX_INTEGRAND=ising needs par(2n+1) in 1..3';stop;endif
   case('stdnorm'); pick=4
   case('mvn'); pick=5
   case('auto')
   case default; write(*,*)'dtt_dmrgg: TTX_INTEGRAND must be auto, host, ising, stdnorm or mvn: ',trim(env); stop
  end select
  if(pick.eq.0)then
   have=.true.; have(5)=allocated(mvn_data%mu).and.mvn_data%n.eq.m
   if(any(n(1:m).ne.n1))have=.false.                  ! the drivers' integrands share one node set over all modes
   ok=have; good=0
   do t=1,NPROBE
    do i=1,m                                          ! ends, centre and scattered interior points
     select case(t)
      case(1); ind(i)=(n(i)+1)/2
      case(2); ind(i)=max(1,min(n(i),(n(i)+1)/2+mod(i,3)-1))
      case(3); ind(i)=1+mod(i,2)*(n(i)-1)
      case default; ind(i)=mod(7*t+3*i+i*i*t,n(i))+1
     end select
    end do
    f=fun(m,ind,n,par)
    do i=1,m; x(i)=par(ind(i)); end do
    do c=1,3; g(c)=0.d0; if(have(c))g(c)=ising_value(c,m,n1,ind,par); end do
    g(4)=0.d0; if(have(4))g(4)=exp(-sum(x**2))
    g(5)=0.d0; if(have(5))g(5)=mvn_pdf(x)
    do c=1,5
     if(.not.ok(c))cycle
     if(abs(f-g(c)).gt.1d-12*abs(g(c)))ok(c)=.false.
     if(abs(g(c)).gt.1d-250)good(c)=good(c)+1
    end do
   end do
   hits=0
   do c=1,5
    if(ok(c).and.good(c).ge.3)then; hits=hits+1; pick=c; endif
   end do
   if(hits.ne.1)pick=0
  end if
  select case(pick)
   case(1:3)
    fid=TTX_FUN_ISING; npar=2*n1+1
    allocate(pcopy(npar)); pcopy(1:2*n1)=par(1:2*n1); pcopy(npar)=dble(pick)
   case(4)
    fid=TTX_FUN_STDNORM; npar=2*n1; allocate(pcopy(npar)); pcopy=par(1:npar)
   case(5)
    if(.not.(allocated(mvn_data%mu).and.mvn_data%n.eq.m))then;write(*,*)'dtt_dmrgg: mvn integrand without mvn_init';stop;endif
    fid=TTX_FUN_MVN; npar=2*n1; allocate(pcopy(npar)); pcopy=par(1:npar)
    allocate(aux(m+m*m+1)); aux(1:m)=mvn_data%mu; aux(m+1:m+m*m)=reshape(mvn_data%inv_cov,[m*m]); aux(m+m*m+1)=mvn_data%det_cov
   case default
    fid=TTX_FUN_HOST; npar=0
  end select
 end subroutine
 double precision function ising_value(id,m,n1,ind,par) result(g)
  ! the Ising-class integrands on the node grid (test_crs_ising.f90:176-218), host arithmetic for the probe only
  integer,intent(in) :: id,m,n1,ind(m)
  double precision,intent(in) :: par(*)
  integer :: i,jj
  double precision :: v,w,vk,wk,a,uij
  a=1.d0
  if(id.ge.2)then
   do i=0,m; uij=1.d0
    do jj=i+1,m; uij=uij*par(ind(jj)); a=a*((uij-1.d0)/(uij+1.d0))**2; end do
   end do
  end if
  v=1.d0;w=1.d0;vk=1.d0;wk=1.d0
  do i=1,m; vk=vk*par(ind(m-i+1)); wk=wk*par(ind(i)); v=v+vk; w=w+wk; end do
  select case(id); case(1);g=2/(v*w); case(2);g=2*a/(v*w); case default;g=2*a; end select
  do i=1,m; g=g*par(n1+ind(i)); end do
 end function
 subroutine ttx_world(wrank,wsize)
  ! this process within the multi-GPU job: TTX_WORLD_RANK / TTX_WORLD_SIZE (default: a single process)
  integer,intent(out) :: wrank,wsize
  character(len=32) :: env
  integer :: stat
  wrank=0; wsize=1
  call get_environment_variable('TTX_WORLD_SIZE',env,status=stat)
  if(stat.eq.0)read(env,*)wsize
  call get_environment_variable('TTX_WORLD_RANK',env,status=stat)
  if(stat.eq.0)read(env,*)wrank
  if(wsize.lt.1.or.wrank.lt.0.or.wrank.ge.wsize)then;write(*,*)'dtt_dmrgg: bad TTX_WORLD_RANK/TTX_WORLD_SIZE: ',wrank,wsize;stop;endif
 end subroutine
 integer(c_int) function ttx_comm_init_shm_env(h) result(rc)
  type(c_ptr),value :: h
  character(len=256) :: nam
  character(kind=c_char) :: cnam(257)
  integer :: stat,i
  call get_environment_variable('TTX_SHM_NAME',nam,status=stat)
  if(stat.ne.0)then       ! default: one name per launcher (the ranks of a job share their parent process), not one for the whole node
   write(nam,'(a,i0)')'ttx_job_',ttx_getppid()
  end if
  do i=1,len_trim(nam); cnam(i)=nam(i:i); end do
  cnam(len_trim(nam)+1)=c_null_char
  rc=ttx_comm_init_shm(h,cnam)
 end function
 integer(c_int) function ttx_comm_init_file(h) result(rc)
  ! RCCL bootstrap without MPI: rank 0 writes the 128-byte unique id to the file TTX_COMM_FILE (written under a temporary
  ! name and renamed, so readers never see a partial file); the other ranks wait for it.  With an MPI build of the
  ! reference this is one mpi_bcast instead (INTEGRATION.md).
  type(c_ptr),value :: h
  integer(c_int8_t),target :: id(128)
  character(len=512) :: fnam
  character(len=540) :: fnam2
  integer :: stat,wrank,wsize,u,tries,ios,fsz
  integer,save :: comm_seq=0
  logical :: there
  call ttx_world(wrank,wsize)
  call get_environment_variable('TTX_COMM_FILE',fnam,status=stat)
  if(stat.ne.0)then;write(*,*)'dtt_dmrgg: TTX_WORLD_SIZE > 1 needs TTX_COMM_FILE (a path all ranks can reach)';stop;endif
  ! One id file per initialisation: the name carries a sequence number that every rank counts the same way (one per dtt_dmrgg
  ! call of the job), so a later call never reads the id of an earlier one; rank 0 removes what a crashed job may have left
  ! under the name before it writes, and removes its own file once every rank has joined (ttx_comm_init returns).
  comm_seq=comm_seq+1
  write(fnam2,'(a,a,i0)')trim(fnam),'.',comm_seq
  if(wrank.eq.0)then
   open(newunit=u,file=trim(fnam2),status='old',iostat=ios); if(ios.eq.0)close(u,status='delete')
   rc=ttx_comm_unique_id(id); if(rc.ne.0)return
   open(newunit=u,file=trim(fnam2)//'.tmp',access='stream',form='unformatted',status='replace')
   write(u)id; close(u)
   call rename(trim(fnam2)//'.tmp',trim(fnam2))
  else
   do tries=1,6000
    inquire(file=trim(fnam2),exist=there,size=fsz)
    if(there.and.fsz.eq.128)exit
    ios=ttx_usleep(100000_c_int32_t)
   end do
   if(.not.(there.and.fsz.eq.128))then;write(*,*)'dtt_dmrgg: no unique id in ',trim(fnam2);stop;endif
   open(newunit=u,file=trim(fnam2),access='stream',form='unformatted',status='old',iostat=ios)
   if(ios.ne.0)then;write(*,*)'dtt_dmrgg: cannot open ',trim(fnam2);stop;endif
   read(u,iostat=ios)id; close(u)
   if(ios.ne.0)then;write(*,*)'dtt_dmrgg: short read of the unique id in ',trim(fnam2);stop;endif
  end if
  rc=ttx_comm_init(h,id)
  if(wrank.eq.0)then
   open(newunit=u,file=trim(fnam2),status='old',iostat=ios); if(ios.eq.0)close(u,status='delete')
  end if
 end function
end module
