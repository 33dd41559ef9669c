! test_ttio.f90 -- small checker of the ttio_lib drop-in: read a stream file, report its shape and the checksum the
! golden fixture tests/golden/ttio_5.txt holds, round it on the device, and write both trains back.
program test_ttio
 use tt_lib
 use ttio_lib
 implicit none
 type(dtt) :: tt,one
 character(len=256) :: fin,fout,fone
 integer :: info,b,i,j,k
 double precision :: s
 call get_command_argument(1,fin); call get_command_argument(2,fout); call get_command_argument(3,fone)
 call read(tt,trim(fin),info)
 write(*,'(a,i3)') 'read info',info
 if(info.ne.0) stop 1
 write(*,'(a,2i4)') 'lm',tt%l,tt%m
 write(*,'(a,16i4)') 'n',tt%n(tt%l:tt%m)
 write(*,'(a,16i4)') 'r',tt%r(tt%l-1:tt%m)
 s=0.d0
 do b=tt%l,tt%m
  do k=1,tt%r(b); do j=1,tt%n(b); do i=1,tt%r(b-1)
   s=s+tt%u(b)%p(i,j,k)*dble(i+2*j+3*k+4*b)
  end do; end do; end do
 end do
 write(*,'(a,e25.17)') 'checksum',s
 write(*,'(a,e25.17)') 'norm',norm(tt)
 call write(tt,trim(fout),info)
 write(*,'(a,i3)') 'write info',info
 one%l=1; one%m=4; one%n(1:4)=(/2,3,4,5/); call ones(one)
 call write(one,trim(fone),info)
 write(*,'(a,i3)') 'write ones info',info
 call read(one,'/nonexistent/file.tt',info)
 write(*,'(a,i3)') 'missing info',info
 call dealloc(tt); call dealloc(one)
end program
