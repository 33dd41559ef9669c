! test_crs_user -- dtt_dmrgg with an integrand that is NOT built into the device engine: the reference's public
! contract is a user callback fun(m,ind,n,par) (lib/dmrgg.f90:18); the drop-in evaluates it on the host
! (ttx_set_integrand_host) while the sweep runs on the GPU.  f(x) = cos(x_1+..+x_D) / (1 + x_1^2+..+x_D^2) on [0,1]^D.
! CLI: D N RANK PIV [MODE]   MODE 0: par and maxrank given; 1: par ABSENT (the callback reads module data, like
! calc_coefficient of test_crs_coscoeff.f90:186); 2: par and maxrank absent (accuracy rule only)
module user_nodes
 implicit none
 double precision,allocatable :: xs(:)
end module

program main
 use tt_lib
 use dmrgg_lib
 use time_lib
 use quad_lib
 use default_lib
 use user_nodes
 implicit none
 include 'mpif.h'
 double precision,parameter :: a=0.d0,b=1.d0
 double precision :: acc
 integer :: mode
 double precision,external :: userfun_nopar
 include 'test_crs_box.inc'
 call readarg(5,mode,0)
 acc=500*epsilon(1.d0)
 allocate(xs(n)); xs=par(1:n)
 select case(mode)
  case(0); call dtt_dmrgg(tt,integrand,par,maxrank=r,accuracy=acc,pivoting=piv,neval=neval,quad=qq)
  case(1); call dtt_dmrgg(tt,userfun_nopar,maxrank=r,accuracy=acc,pivoting=piv,neval=neval,quad=qq)
  case default; call dtt_dmrgg(tt,userfun_nopar,accuracy=acc,pivoting=piv,neval=neval,quad=qq)
 end select
 t2=timef()
 write(*,'(a,i12,a,e12.4,a)') '...with',neval,' evaluations completed in ',t2-t1,' sec.'
 val=dtt_quad(tt,qq)
 write(*,'(a,e50.40)') 'computed value:',val
 write(*,'(a)') 'Good bye.'
 call dealloc(tt)
 call mpi_finalize(info)
end program

double precision function integrand(m,ind,n,par) result(f)
 implicit none
 integer,intent(in) :: m
 integer,intent(in) :: ind(m),n(m)
 double precision,intent(in) :: par(*)
 double precision :: s1,s2
 integer :: i
 s1=0.d0; s2=0.d0
 do i=1,m; s1=s1+par(ind(i)); s2=s2+par(ind(i))*par(ind(i)); end do
 f=cos(s1)/(1.d0+s2)
end function

double precision function userfun_nopar(m,ind,n,par) result(f)
 use user_nodes
 implicit none
 integer,intent(in) :: m
 integer,intent(in) :: ind(m),n(m)
 double precision,intent(in),optional :: par(*)
 double precision :: s1,s2
 integer :: i
 s1=0.d0; s2=0.d0
 do i=1,m; s1=s1+xs(ind(i)); s2=s2+xs(ind(i))*xs(ind(i)); end do
 f=cos(s1)/(1.d0+s2)
end function
