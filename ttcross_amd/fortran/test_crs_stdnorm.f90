! test_crs_stdnorm -- integral of exp(-|x|^2) over [-10,10]^D (= sqrt(pi)^D); reference CLI: D N RANK PIV
program main
 use tt_lib
 use dmrgg_lib
 use time_lib
 use quad_lib
 use default_lib
 implicit none
 include 'mpif.h'
 double precision,parameter :: a=-10.d0,b=10.d0
 double precision :: acc,tru
 include 'test_crs_box.inc'
 acc=5*epsilon(1.d0)
 tru=sqrt(3.141592653589793238d0)**d
 call dtt_dmrgg(tt,integrand,par,maxrank=r,accuracy=acc,pivoting=piv,neval=neval,quad=qq,tru=tru)
 t2=timef()
 if(me.eq.0)write(*,'(a,i12,a,e12.4,a)') '...with',neval,' evaluations completed in ',t2-t1,' sec.'
 val=dtt_quad(tt,qq)                          ! collective; the report is rank 0's
 if(me.ne.0)then
  call dealloc(tt); call mpi_finalize(info); stop
 end if
 write(*,'(a,e50.40)') 'computed value:',val
 write(*,'(a,e50.40)') 'analytic value:',tru
 write(*,'(a,f7.2)') 'correct digits:',-dlog(dabs(1.d0-val/tru))/dlog(10.d0)
 write(*,'(a)') 'Good bye.'
 call dealloc(tt)
 call mpi_finalize(info)
end program

double precision function integrand(m,ind,n,par) result(f)
 implicit none
 integer,intent(in) :: m
 integer,intent(in) :: ind(m),n(m)
 double precision,intent(inout),optional :: par(*)
 double precision :: x(m)
 integer :: i
 do i=1,m; x(i)=par(ind(i)); end do
 f=exp(-sum(x**2))
end function
