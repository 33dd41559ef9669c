! test_crs_chf -- characteristic function of the basket average under a correlated multivariate normal:
! TT cross of the density without a quadrature argument, then 32 complex rank-1 quadratures of the device train
! (the pipeline of the reference's test_crs_chf.f90:104-168); CLI: D N RANK PIV
program main
 use tt_lib
 use dmrgg_lib
 use time_lib
 use quad_lib
 use default_lib
 use mvn_pdf_mod
 implicit none
 include 'mpif.h'
 double precision,parameter :: a=0.525170,b=8.525170,pi=3.14159265358979323846d0
 double precision :: acc,omega
 type(ztt) :: tz,wq
 double complex :: ans(32)
 double complex,allocatable :: wc(:)
 integer :: k,p
 include 'test_crs_box.inc'
 acc=500*epsilon(1.d0)
 call mvn_init(d,0.d0,1.d0)
 call dtt_dmrgg(tt,integrand,par,maxrank=r,accuracy=acc,pivoting=piv,neval=neval)
 t2=timef()
 write(*,'(a,i12,a,e12.4,a)') '...with',neval,' evaluations completed in ',t2-t1,' sec.'
 tz=tt
 allocate(wc(n))
 wq%l=1; wq%m=d; wq%n=n; wq%r=1; call alloc(wq)
 do k=0,31
  omega=k*pi/(300.d0-0.d0)
  do p=1,n; wc(p)=exp((0.d0,1.d0)*omega*exp(par(p))/dble(d)); end do
  do i=1,d; wq%u(i)%p(1,:,1)=dcmplx(par(n+1:2*n),0.d0)*wc; end do
  ans(k+1)=ztt_quad(tz,wq)
 end do
 do k=0,31
  write(*,'(a,i3,2e26.17)') 'computed value:',k,dble(ans(k+1)),dimag(ans(k+1))
 end do
 write(*,'(a)') 'Good bye.'
 call dealloc(wq); call dealloc(tz); call dealloc(tt)
 call mpi_finalize(info)
end program

double precision function integrand(m,ind,n,par) result(f)
 use mvn_pdf_mod
 implicit none
 integer,intent(in) :: m
 integer,intent(in) :: ind(m),n(m)
 double precision,intent(inout),optional :: par(*)
 double precision :: x(m)
 integer :: i
 do i=1,m; x(i)=par(ind(i)); end do
 f=mvn_pdf(x)
end function
