! blas1.f90 -- the BLAS-1 symbols the reference's drivers call directly (test_crs_ising.f90:103,131,138,142:
! dscal, dcopy), for builds that do not link a BLAS.  Same interfaces as the netlib routines; a driver that links
! -lblas can simply leave this object out.
subroutine dscal(n,da,dx,incx)
 integer :: n,incx,i
 double precision :: da,dx(*)
 if(n.le.0.or.incx.le.0)return
 do i=1,1+(n-1)*incx,incx; dx(i)=da*dx(i); end do
end subroutine
subroutine dcopy(n,dx,incx,dy,incy)
 integer :: n,incx,incy,i,ix,iy
 double precision :: dx(*),dy(*)
 if(n.le.0)return
 ix=1; iy=1
 if(incx.lt.0)ix=(-n+1)*incx+1
 if(incy.lt.0)iy=(-n+1)*incy+1
 do i=1,n; dy(iy)=dx(ix); ix=ix+incx; iy=iy+incy; end do
end subroutine
