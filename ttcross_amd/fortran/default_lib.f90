! default_lib / time_lib / quad_lib -- host helpers the drivers use (lib/default.f90:10-97, lib/timef.f90:25,
! lib/quad.f90:97-131), restated.
module default_lib
 implicit none
 interface default; module procedure default_i,default_d; end interface
 interface readarg; module procedure readarg_i,readarg_a; end interface
contains
 integer function default_i(def,opt) result(f)
  integer,intent(in) :: def; integer,intent(in),optional :: opt
  f=def; if(present(opt))f=opt
 end function
 double precision function default_d(def,opt) result(f)
  double precision,intent(in) :: def; double precision,intent(in),optional :: opt
  f=def; if(present(opt))f=opt
 end function
 subroutine readarg_i(pos,arg,opt)
  integer,intent(in) :: pos; integer,intent(out) :: arg; integer,intent(in),optional :: opt
  character(len=128) :: str
  if(pos.le.0)then;write(*,*)'readarg_i: illegal pos:',pos;stop;endif
  call get_command_argument(pos,str)
  if(present(opt).and.str.eq.' ')then; arg=opt; else; read(str,*)arg; endif
 end subroutine
 subroutine readarg_a(pos,arg,opt)
  integer,intent(in) :: pos; character,intent(out) :: arg; character,intent(in),optional :: opt
  character(len=128) :: str
  if(pos.le.0)then;write(*,*)'readarg_a: illegal pos:',pos;stop;endif
  call get_command_argument(pos,str)
  if(present(opt).and.str.eq.' ')then; arg=opt; else; read(str,'(a1)')arg; endif
 end subroutine
 subroutine share(first,last,own)
  ! bonds first..last over nproc ranks; nproc comes from the (single-process) MPI stub
  integer,intent(in) :: first,last; integer,intent(out) :: own(0:)
  integer :: p,nproc
  nproc=ubound(own,1)
  own(0)=first
  do p=1,nproc-1; own(p)=first+int(dble(last-first+1)*dble(p)/nproc); enddo
  own(nproc)=last+1
 end subroutine
end module

module time_lib
 implicit none
contains
 double precision function timef()
  integer(8) :: c,r
  call system_clock(count=c,count_rate=r)
  timef=dble(c)/dble(r)
 end function
end module

module quad_lib
 implicit none
 double precision,parameter,private :: tpi=6.28318530717958647692528676655900576839433879875d0
contains
 subroutine lgwt(n,x,w)
  ! Gauss-Legendre nodes and weights on [-1,1] by Newton iteration on P_n
  integer,intent(in) :: n
  double precision,intent(out) :: x(n),w(n)
  double precision :: small,p1,p2,p3,pp,z,z1
  integer :: i,j
  small=5*epsilon(1.d0)
  do i=1,(n+1)/2
   z=dcos((tpi*(4*i-1))/(8*n+4))
   do
    p1=1.0d0; p2=0.0d0
    do j=1,n
     p3=p2; p2=p1
     p1=((2*j-1)*z*p2-(j-1)*p3)/j
    end do
    pp=n*(z*p1-p2)/(z*z-1)
    z1=z
    z=z1-p1/pp
    if(dabs(z-z1).le.small)exit
   end do
   x(i)=-z; x(n+1-i)=z
   w(i)=2.d0/((1-z*z)*pp*pp); w(n+1-i)=w(i)
  end do
 end subroutine
end module
