! test_crs_mvn -- mass of a correlated multivariate normal over a box; reference CLI: D N RANK PIV
program main
 use tt_lib
 use dmrgg_lib
 use time_lib
 use quad_lib
 use default_lib
 use mvn_pdf_mod
 implicit none
 include 'mpif.h'
 double precision,parameter :: a=0.525170,b=8.525170     ! single-precision literals, as in the reference driver
 double precision :: acc,tru
 include 'test_crs_box.inc'
 acc=500*epsilon(1.d0)
 tru=1.d0
 call mvn_init(d,0.d0,1.d0)
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
