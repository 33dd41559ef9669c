! Timing driver (bench infrastructure, NOT product code): the GENUINE reference's tt_lib utilities -- dtt_ort (lib/tt.f90:130-198)
! and dtt_svd (lib/tt.f90:307-368) -- timed on a train read with the reference's own dtt_read (lib/ttio.f90:196-297) from the stream
! file that bench.py wrote with ttx_write.  Built by `make -C oracle ref` into oracle/_ref/ref_tt_timing from the reference's sources
! where they lie; used only by bench.py's cpu_baseline leg of `--workload ort|svd[_d64]`.
!    ref_tt_timing FILE ort|svd TOL SECONDS     -> "calls <k> median_ms <t> ranks_max <r>"
program ref_tt_timing
 use tt_lib
 use ttio_lib
 implicit none
 type(dtt) :: tt,t1
 character(len=512) :: fnam,op,arg
 integer :: info,k,calls,rmax
 double precision :: tol,budget,t0,t1s,tot,ts(4096),tmp
 integer(kind=8) :: c0,c1,rate
 call get_command_argument(1,fnam); call get_command_argument(2,op)
 call get_command_argument(3,arg); read(arg,*)tol
 call get_command_argument(4,arg); read(arg,*)budget
 call read(tt,trim(fnam),info)
 if(info.ne.0)then; write(*,*)'ref_tt_timing: cannot read ',trim(fnam),info; stop 1; endif
 calls=0; tot=0.d0; rmax=0
 do while(tot.lt.budget .and. calls.lt.4096)
  t1=tt
  call system_clock(c0,rate)
  if(trim(op).eq.'ort')then
   call ort(t1)
  else
   call svd(t1,tol)
  end if
  call system_clock(c1)
  calls=calls+1; ts(calls)=dble(c1-c0)/dble(rate); tot=tot+ts(calls)
  rmax=maxval(t1%r(t1%l-1:t1%m))
  call dealloc(t1)
 end do
 ! median by insertion sort
 do k=2,calls
  tmp=ts(k); info=k-1
  do while(info.ge.1)
   if(ts(info).le.tmp)exit
   ts(info+1)=ts(info); info=info-1
  end do
  ts(info+1)=tmp
 end do
 write(*,'(a,i6,a,f14.6,a,i5)')'calls ',calls,' median_ms ',1.d3*ts((calls+1)/2),' ranks_max ',rmax
end program
