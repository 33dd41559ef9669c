! Fixture generator (test infrastructure): calls the GENUINE reference modules (compiled into oracle/_ref by
! `make -C oracle ref`) -- dtt_dmrgg on Ising C_m followed by dtt_accchk (lib/dmrgg.f90:1081) -- and prints
! the four error measures and the worst multi-index.  The integrand below is this repository's own restatement.
program ref_accchk
 use tt_lib
 use dmrgg_lib
 use quad_lib
 use default_lib
 implicit none
 include 'mpif.h'
 type(dtt) :: tt,qq
 integer :: i,m,n,r,piv,nlot,info,pv(tt_size)
 integer(kind=8) :: neval
 double precision :: acc,einf,efro,ainf,afro,sc
 double precision,allocatable :: par(:)
 double precision,external :: isingc
 call readarg(1,m,6); call readarg(2,n,33); call readarg(3,r,12); call readarg(4,piv,2); call readarg(5,nlot,2000)
 call mpi_init(info)
 allocate(par(2*n+1)); par(2*n+1)=1.d0
 call lgwt(n,par(1),par(n+1))
 par(n+1:2*n)=0.5d0*par(n+1:2*n); par(1:n)=(par(1:n)+1.d0)/2
 sc=dble(n/2); par(n+1:2*n)=sc*par(n+1:2*n)
 qq%l=1;qq%m=m-1;qq%n=n;qq%r=1;call alloc(qq)
 do i=1,m-1; qq%u(i)%p=1.d0/sc; end do
 acc=500*epsilon(1.d0)
 tt%l=1;tt%m=m-1;tt%n=n;tt%r=1;call alloc(tt)
 call dtt_dmrgg(tt,isingc,par,maxrank=r,accuracy=acc,pivoting=piv,neval=neval,quad=qq)
 call dtt_accchk(nlot,tt,einf,efro,ainf,afro,isingc,par,pv)
 write(*,'(a,4e25.17)') 'accchk ',einf,efro,ainf,afro
 write(*,'(a,64i4)') 'pivot ',pv(1:m-1)
 call mpi_finalize(info)
end program
double precision function isingc(m,ind,n,par) result(f)
 implicit none
 integer,intent(in) :: m
 integer,intent(in) :: ind(m),n(m)
 double precision,intent(inout),optional :: par(*)
 integer :: i
 double precision :: v,w,vk,wk
 v=1.d0;w=1.d0;vk=1.d0;wk=1.d0
 do i=1,m; vk=vk*par(ind(m-i+1)); wk=wk*par(ind(i)); v=v+vk; w=w+wk; end do
 f=2*(1.d0/(v*w))
 do i=1,m; f=f*par(n(1)+ind(i)); end do
end function
