! Fixture generator (test infrastructure): the GENUINE reference's dtt_write (lib/ttio.f90:29-108) applied to a small
! train with closed-form cores u_b(i,j,k) = (3*i + 5*j + 7*k + 11*b) / 16 (exact in binary), and its dtt_read
! (lib/ttio.f90:196-297) applied to a file named on the command line (prints a checksum line).
program ref_ttio
 use tt_lib
 use ttio_lib
 implicit none
 type(dtt) :: tt,t2
 integer :: b,i,j,k,info
 integer,parameter :: d=5
 integer,parameter :: nn(5)=(/3,4,2,5,3/), rr(0:5)=(/1,2,3,2,2,1/)
 character(len=256) :: fin,fout
 double precision :: s
 call get_command_argument(1,fout)
 call get_command_argument(2,fin)
 tt%l=1; tt%m=d; tt%n(1:d)=nn; tt%r(0:d)=rr; call alloc(tt)
 do b=1,d
  do k=1,rr(b); do j=1,nn(b); do i=1,rr(b-1)
   tt%u(b)%p(i,j,k)=dble(3*i+5*j+7*k+11*b)/16.d0
  end do; end do; end do
 end do
 call write(tt,trim(fout),info)
 write(*,'(a,i3)') 'write info',info
 if(len_trim(fin).gt.0)then
  call read(t2,trim(fin),info)
  write(*,'(a,i3)') 'read info',info
  write(*,'(a,2i4)') 'lm',t2%l,t2%m
  write(*,'(a,16i4)') 'n',t2%n(t2%l:t2%m)
  write(*,'(a,16i4)') 'r',t2%r(t2%l-1:t2%m)
  s=0.d0
  do b=t2%l,t2%m
   do k=1,t2%r(b); do j=1,t2%n(b); do i=1,t2%r(b-1)
    s=s+t2%u(b)%p(i,j,k)*dble(i+2*j+3*k+4*b)
   end do; end do; end do
  end do
  write(*,'(a,e25.17)') 'checksum',s
 end if
end program
