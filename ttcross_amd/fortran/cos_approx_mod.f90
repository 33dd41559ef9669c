! cos_approx_mod.f90 -- drop-in for the reference's cos_approx_mod (lib/cos_approx.f90): the COS-method density from the
! characteristic-function values phi(omega_k), omega_k = k pi / (b - a), that the 32 ztt_quad calls of the fork's drivers
! produce (test_crs_pdf.f90:153-190):
!     f(x) ~ sum_{k=0}^{n-1} ' c_k cos(omega_k (x - a)),   c_k = 2/(b-a) Re( phi_k exp(-i omega_k a) ),   ' = first term halved.
! Host arithmetic on a handful of numbers (consumer of the device results); same names and argument lists as the reference.
module cos_approx_mod
 implicit none
 private
 public :: cos_approximate, cos_approximate_array
contains
 pure subroutine cos_coefficients(phis,a,b,n,c,w)
  double complex,intent(in) :: phis(:)
  double precision,intent(in) :: a,b
  integer,intent(in) :: n
  double precision,intent(out) :: c(n),w(n)
  double precision,parameter :: pi=3.1415926535897932384626433832795d0
  integer :: k
  do k=1,n
   w(k)=(k-1)*(pi/(b-a))
   c(k)=2.d0/(b-a)*dble(phis(k)*exp(-1.d0*(0.d0,1.d0)*w(k)*a))
  end do
  c(1)=c(1)/2.d0
 end subroutine
 function cos_approximate(x,phis,lower_bound,upper_bound,n_terms) result(pdf_val)
  double precision,intent(in) :: x
  double complex,intent(in) :: phis(:)
  double precision,intent(in) :: lower_bound,upper_bound
  integer,intent(in),optional :: n_terms
  double precision :: pdf_val
  double precision,allocatable :: c(:),w(:)
  integer :: n,k
  n=size(phis); if(present(n_terms))n=n_terms
  pdf_val=0.d0
  if(n.gt.size(phis))then; print *,'Error: n_terms exceeds the size of phis.'; return; endif
  allocate(c(n),w(n)); call cos_coefficients(phis,lower_bound,upper_bound,n,c,w)
  do k=1,n; pdf_val=pdf_val+c(k)*cos(w(k)*(x-lower_bound)); end do
 end function
 subroutine cos_approximate_array(xs,phis,lower_bound,upper_bound,n_terms,pdf_vals)
  double precision,intent(in) :: xs(:)
  double complex,intent(in) :: phis(:)
  double precision,intent(in) :: lower_bound,upper_bound
  integer,intent(in),optional :: n_terms
  double precision,intent(out) :: pdf_vals(size(xs))
  double precision,allocatable :: c(:),w(:)
  integer :: n,k,i
  n=size(phis); if(present(n_terms))n=n_terms
  pdf_vals=0.d0
  if(n.gt.size(phis))then; print *,'Error: n_terms exceeds the size of phis.'; return; endif
  allocate(c(n),w(n)); call cos_coefficients(phis,lower_bound,upper_bound,n,c,w)
  do k=1,n
   do i=1,size(xs); pdf_vals(i)=pdf_vals(i)+c(k)*cos(w(k)*(xs(i)-lower_bound)); end do
  end do
 end subroutine
end module
