! test_crs_ising -- Ising-class integrals C_m, D_m, E_m by TT cross interpolation on the MI355X engine.
! Same command line and report as the reference driver (test_crs_ising.f90):  KIND INDEX N RANK PIV
program main
 use tt_lib
 use dmrgg_lib
 use time_lib
 use quad_lib
 use default_lib
 implicit none
 include 'mpif.h'
 type(dtt) :: tt,qq
 integer :: i,m,n,r,piv,info,nproc,me,adj
 integer(kind=8) :: neval
 double precision :: t1,t2,acc,val,tru,scal
 double precision,allocatable :: par(:)
 character(len=1) :: a
 logical :: rescale
 double precision,external :: dfunc_ising_discr
 call readarg(1,a,'c'); call readarg(2,m,6); call readarg(3,n,65); call readarg(4,r,20); call readarg(5,piv,1)
 call mpi_init(info); call mpi_comm_size(MPI_COMM_WORLD,nproc,info); call mpi_comm_rank(MPI_COMM_WORLD,me,info)
 adj=0; if(mod(n,2).eq.0)then; n=n+1; adj=1; endif
 if(me.eq.0)call banner()
 acc=500*epsilon(1.d0)
 allocate(par(2*n+1))
 select case(a)
  case('c','C'); par(2*n+1)=1.d0
  case('d','D'); par(2*n+1)=2.d0
  case('e','E'); par(2*n+1)=3.d0
  case default; write(*,*)'unknown integral type:',a; stop
 end select
 tru=ising_value(a,m)
 call lgwt(n,par(1),par(n+1))
 par(n+1:2*n)=0.5d0*par(n+1:2*n)            ! a probability measure on [0,1]
 par(1:n)=(par(1:n)+1.d0)/2
 qq%l=1; qq%m=m-1; qq%n=n; qq%r=1; call alloc(qq)
 rescale=(a.eq.'d'.or.a.eq.'D'.or.a.eq.'e'.or.a.eq.'E').and.(m.ge.10)
 scal=dble(n/2)
 if(rescale)then; par(n+1:2*n)=5.d0*scal*par(n+1:2*n); else; par(n+1:2*n)=scal*par(n+1:2*n); endif
 do i=1,m-1; qq%u(i)%p=1.d0/scal; end do
 t1=timef()
 tt%l=1; tt%m=m-1; tt%n=n; tt%r=1; call alloc(tt)
 if(tru.eq.0.d0)then
  call dtt_dmrgg(tt,dfunc_ising_discr,par,maxrank=r,accuracy=acc,pivoting=piv,neval=neval,quad=qq)
 else
  call dtt_dmrgg(tt,dfunc_ising_discr,par,maxrank=r,accuracy=acc,pivoting=piv,neval=neval,quad=qq,tru=tru)
 endif
 t2=timef()
 if(me.eq.0)write(*,'(a,i12,a,e12.4,a)') '...with',neval,' evaluations completed in ',t2-t1,' sec.'
 val=dtt_quad(tt,qq)                          ! collective: every process takes part, the report is rank 0's (test_crs_ising.f90:158-169)
 if(me.ne.0)then
  call dealloc(tt); call mpi_finalize(info); stop
 end if
 if(rescale)then
  write(*,'(a,e50.40,a,i4,a)') 'computed value:',val,' / (5**',m-1,')'
 else
  write(*,'(a,e50.40)') 'computed value:',val
 end if
 if(tru.ne.0.d0)then
  write(*,'(a,e50.40)') 'analytic value:',tru
  write(*,'(a,f7.2)')   'correct digits:',-dlog(dabs(1.d0-val/tru))/dlog(10.d0)
 end if
 write(*,'(a)')'Good bye.'
 call dealloc(tt)
 call mpi_finalize(info)
contains
 subroutine banner()
  ! the reference driver's header, line for line (values only differ in the engine line)
  character(len=9),parameter :: lab(6) = ['dimension','quadratur','TT ranks ','pivoting ','MPI procs','sizeof(d)']
  integer :: val(6),q
  val = [m, n, r, piv, nproc, storage_size(1.d0)]
  write(*,'(a)') 'Hi, this is TT cross interpolation computing Ising integral...'
  write(*,'(3x,a,a10)') 'integral :',a
  do q = 1, 6
   if(q == 2 .and. adj /= 0) then
    write(*,'(3x,a,a,i10,a)') lab(q),':',val(q),' (adjusted)'
   else
    write(*,'(3x,a,a,i10)') lab(q),':',val(q)
   end if
   if(q == 5) write(*,'(3x,a,a10)') 'engine   :','MI355X HIP'
  end do
  write(*,'(3x,a,e10.3)') 'epsilon  :',epsilon(1.d0)
 end subroutine
 double precision function ising_value(a,m) result(t)
  ! Bailey, Borwein & Crandall, "Integrals of the Ising class" (2006), rounded to double
  character(len=1),intent(in) :: a
  integer,intent(in) :: m
  t=0.d0
  select case(a)
  case('c','C')
   select case(m)
    case(2); t=1.d0
    case(3); t=0.78130241289648629687d0
    case(4); t=0.70119986017642999982d0
    case(5); t=0.66575980019993742832d0
    case(6); t=0.64863420903100707526d0
    case(8); t=0.63548402675916322614d0
    case(16); t=0.63050394617323726351d0
    case(32); t=0.63047350420733980638d0
    case(64); t=0.63047350337438679649d0
    case(128,256,512,1024); t=0.63047350337438679612d0
   end select
  case('d','D')
   select case(m)
    case(2); t=1.d0/3
    case(5); t=0.0024846057623403154800d0
    case(6); t=0.00048914170018803477510d0
   end select
  case('e','E')
   select case(m)
    case(5); t=0.0034936537117295217407d0
    case(6); t=0.00068783287182640943700d0
   end select
  end select
 end function
end program

double precision function dfunc_ising_discr(m,ind,n,par) result(f)
 ! the user integrand of the Ising driver: f = 2 a b prod(w), a = prod_{i<j}((u_ij-1)/(u_ij+1))^2 (D,E),
 ! b = 1/((1+sum of suffix products)(1+sum of prefix products)) (C,D).  Only used on the host to identify
 ! the integrand; the sweep evaluates the device version.
 implicit none
 integer,intent(in) :: m
 integer,intent(in) :: ind(m),n(m)
 double precision,intent(inout),optional :: par(*)
 integer :: i,j,id
 double precision :: uij,a,b,v,w,wk,vk
 id=int(par(2*n(1)+1)); a=1.d0; b=1.d0
 if(id.ge.2)then
  do i=0,m
   uij=1.d0
   do j=i+1,m; uij=uij*par(ind(j)); a=a*((uij-1.d0)/(uij+1.d0))**2; end do
  end do
 end if
 if(id.le.2)then
  v=1.d0;w=1.d0;vk=1.d0;wk=1.d0
  do i=1,m; vk=vk*par(ind(m-i+1)); wk=wk*par(ind(i)); v=v+vk; w=w+wk; end do
  b=1.d0/(v*w)
 end if
 f=2*a*b
 do i=1,m; f=f*par(n(1)+ind(i)); end do
end function
