! mvn_pdf_mod -- same public names as lib/mvn_pdf.f90 (mvn_data, mvn_init, mvn_pdf); the inverse and the
! determinant of the test covariance come from an in-module LU (the reference calls LAPACK dgetrf/dgetri).
module mvn_pdf_mod
 implicit none
 type mvn_data_t
  integer :: n
  real(8),allocatable :: mu(:)
  real(8),allocatable :: inv_cov(:,:)
  real(8) :: det_cov
 end type
 type(mvn_data_t),save :: mvn_data
contains
 subroutine mvn_init(n,r,T)
  integer,intent(in) :: n
  real(8),intent(in) :: r,T
  real(8),parameter :: sigma=0.4d0,corr=0.5d0
  real(8) :: cov(n,n),a(n,n),x(n),X0,rp,tt
  integer :: i,j,k,c,piv,ipiv(n)
  X0=log(100.0d0)
  if(allocated(mvn_data%mu))deallocate(mvn_data%mu,mvn_data%inv_cov)
  allocate(mvn_data%mu(n),mvn_data%inv_cov(n,n))
  mvn_data%n=n
  mvn_data%mu=X0+(r-0.5d0*sigma**2)*T
  do i=1,n; do j=1,n
   if(i.eq.j)then; cov(i,j)=sigma*sigma; else; cov(i,j)=sigma*corr*sigma; endif
   cov(i,j)=cov(i,j)*T
  end do; end do
  a=cov
  do k=1,n
   piv=k
   do i=k+1,n; if(abs(a(i,k)).gt.abs(a(piv,k)))piv=i; end do
   ipiv(k)=piv
   if(piv.ne.k)then; do j=1,n; tt=a(k,j);a(k,j)=a(piv,j);a(piv,j)=tt; end do; endif
   rp=1.d0/a(k,k)
   a(k+1:n,k)=a(k+1:n,k)*rp
   do j=k+1,n; a(k+1:n,j)=a(k+1:n,j)-a(k+1:n,k)*a(k,j); end do
  end do
  mvn_data%det_cov=1.d0
  do i=1,n
   if(ipiv(i).ne.i)mvn_data%det_cov=-mvn_data%det_cov
   mvn_data%det_cov=mvn_data%det_cov*a(i,i)
  end do
  do c=1,n
   x=0.d0; x(c)=1.d0
   do k=1,n; if(ipiv(k).ne.k)then; tt=x(k);x(k)=x(ipiv(k));x(ipiv(k))=tt; endif; end do
   do k=1,n; x(k+1:n)=x(k+1:n)-a(k+1:n,k)*x(k); end do
   do k=n,1,-1; x(k)=x(k)/a(k,k); x(1:k-1)=x(1:k-1)-a(1:k-1,k)*x(k); end do
   mvn_data%inv_cov(:,c)=x
  end do
 end subroutine
 function mvn_pdf(x) result(pdf)
  real(8),intent(in) :: x(:)
  real(8) :: pdf,ex,diff(size(x))
  real(8),parameter :: pi=3.141592653589793d0
  integer :: i,j,n
  n=mvn_data%n
  diff=x-mvn_data%mu
  ex=0.d0
  do i=1,n; do j=1,n; ex=ex+diff(i)*mvn_data%inv_cov(i,j)*diff(j); end do; end do
  pdf=exp(-0.5d0*ex)/sqrt((2.0d0*pi)**n*mvn_data%det_cov)
 end function
end module
