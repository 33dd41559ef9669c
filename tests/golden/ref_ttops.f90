! Fixture generator (test infrastructure): the GENUINE reference's tt_lib utilities (dtt_ort lib/tt.f90:130,
! dtt_svd :307, dtt_norm :1074, dtt_dot :1155, dtt_ijk :630) applied to the TT that dtt_dmrgg builds for Ising C_m.
program ref_ttops
 use tt_lib
 use dmrgg_lib
 use quad_lib
 use default_lib
 implicit none
 include 'mpif.h'
 type(dtt) :: tt,t1,t2,t3
 integer :: i,m,n,r,piv,info,ind(64),k
 integer(kind=8) :: neval
 double precision :: acc,sc,tol
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
 write(*,'(a,64i4)') 'ranks0 ',tt%r(0:m-1)
 write(*,'(a,e25.17)') 'norm0 ',norm(tt)
 write(*,'(a,e25.17)') 'dot00 ',dot_product(tt,tt)
 t1=tt; call ort(t1)
 write(*,'(a,64i4)') 'ranks_ort ',t1%r(0:m-1)
 write(*,'(a,e25.17)') 'norm_ort ',norm(t1)
 do tol=1,3
  t2=tt
  if(tol.eq.1)call svd(t2,1.d-4)
  if(tol.eq.2)call svd(t2,1.d-8)
  if(tol.eq.3)call svd(t2,1.d-12,5)
  write(*,'(a,i2,64i4)') 'ranks_svd',int(tol),t2%r(0:m-1)
  write(*,'(a,i2,e25.17)') 'norm_svd',int(tol),norm(t2)
  write(*,'(a,i2,e25.17)') 'dot_svd',int(tol),dot_product(tt,t2)
  do k=1,4
   do i=1,m-1; ind(i)=mod(5*k+3*i+i*i*k,n)+1; end do
   write(*,'(a,2i2,2e25.17)') 'elem',int(tol),k,tijk(tt,ind(1:m-1)),tijk(t2,ind(1:m-1))
  end do
  call dealloc(t2)
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
