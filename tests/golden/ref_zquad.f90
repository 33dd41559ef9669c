! Fixture generator (test infrastructure): the GENUINE reference's ztt_quad (lib/dmrgg.f90:1418-1523) on the TT that
! dtt_dmrgg builds for Ising C_m, with the complex weights of the fork's characteristic-function driver
! (w_p * exp(i*omega*exp(x_p)/d), omega = k*pi/300, k = 0..7).
program ref_zquad
 use tt_lib
 use dmrgg_lib
 use quad_lib
 use default_lib
 implicit none
 include 'mpif.h'
 type(dtt) :: tt
 type(ztt) :: tz,qq
 integer :: i,m,n,r,piv,info,k,p
 integer(kind=8) :: neval
 double precision :: acc,sc,omega
 double precision,parameter :: pi=3.14159265358979323846d0
 double complex :: ans
 double precision,allocatable :: par(:)
 double precision,external :: isingc
 call readarg(1,m,6); call readarg(2,n,33); call readarg(3,r,12); call readarg(4,piv,2)
 call mpi_init(info)
 allocate(par(2*n+1)); par(2*n+1)=1.d0
 call lgwt(n,par(1),par(n+1))
 par(n+1:2*n)=0.5d0*par(n+1:2*n); par(1:n)=(par(1:n)+1.d0)/2
 sc=dble(n/2); par(n+1:2*n)=sc*par(n+1:2*n)
 acc=500*epsilon(1.d0)
 tt%l=1;tt%m=m-1;tt%n=n;tt%r=1;call alloc(tt)
 call dtt_dmrgg(tt,isingc,par,maxrank=r,accuracy=acc,pivoting=piv,neval=neval)
 tz=tt
 qq%l=1;qq%m=m-1;qq%n=n;qq%r=1;call alloc(qq)
 do k=0,7
  omega=k*pi/300.d0
  do i=1,m-1
   do p=1,n
    qq%u(i)%p(1,p,1)=dcmplx(1.d0/sc,0.d0)*exp((0.d0,1.d0)*omega*exp(par(p))/dble(m-1))
   end do
  end do
  ans=ztt_quad(tz,qq)
  write(*,'(a,i3,2e26.17)') 'zquad',k,dble(ans),dimag(ans)
 end do
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
