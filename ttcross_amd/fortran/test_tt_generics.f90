! test_tt_generics -- the host-side generics of the drop-in tt_lib (lib/tt.f90:54-124) and the ownership rules of the
! hidden device handle: b = a is a deep copy (lib/tt.f90:1012-1020), so dealloc(a); dealloc(b) release disjoint
! storage; + and * build host trains that the device utilities stage on demand.
program main
 use tt_lib
 use mat_lib
 use cos_approx_mod
 use utils
 implicit none
 double complex :: phis(4)
 double precision :: pdf(3)
 character(len=256) :: h5name
 type(dtt) :: a,b,c
 double precision :: s,nrm,nb,e(1)
 double precision,pointer :: u(:,:),sv(:),v(:,:)
 double precision :: mat(3,3),inv(3,3)
 integer :: k
 a%l=1; a%m=4; a%n(1:4)=[2,3,4,5]; call ones(a)
 do k=1,4; a%u(k)%p=0.5d0*k; end do
 b=a                                   ! deep copy
 b%u(2)%p=7.d0
 write(*,'(a,2f8.3)') 'copy_independent ',a%u(2)%p(1,1,1),b%u(2)%p(1,1,1)
 write(*,'(a,f12.4)') 'numel ',numel(a)
 write(*,'(a,i6)') 'memory ',memory(a)
 write(*,'(a,f14.6)') 'sumall ',sumall(a)
 c=a+a
 write(*,'(a,5i3)') 'plus_ranks ',c%r(0:4)
 write(*,'(a,f14.6)') 'plus_sumall ',sumall(c)
 c=3.d0*a
 write(*,'(a,f14.6)') 'mul_sumall ',sumall(c)
 write(*,'(a,f14.6)') 'tijk ',tijk(a,[1,2,3,4])
 call elem(a,[2,3,4,5],e)
 write(*,'(a,f14.6)') 'elem ',e(1)
 write(*,'(a,f14.6)') 'value0 ',value(a,0.3d0)
 nrm=norm(a)                           ! staged on the device for the call
 write(*,'(a,f14.6)') 'norm ',nrm
 write(*,'(a,f14.6)') 'lognrm ',lognrm(a)
 write(*,'(a,f14.6)') 'dot ',dot_product(a,a)
 c=a+a
 call svd(c,1.d-12)                    ! rounding of a+a on the device: ranks back to 1
 write(*,'(a,5i3)') 'svd_ranks ',c%r(0:4)
 nb=norm(c)
 write(*,'(a,f14.6)') 'svd_norm ',nb
 call zeros(b)
 write(*,'(a,f14.6)') 'zeros_sumall ',sumall(b)
 call copy(a,b)
 write(*,'(a,f14.6)') 'copy_sumall ',sumall(b)
 call say(a)
 ! mat_lib
 mat=reshape([4.d0,1.d0,0.d0, 1.d0,3.d0,1.d0, 0.d0,1.d0,2.d0],[3,3])
 inv=mat; call matinv(inv,alg='t')
 write(*,'(a,e12.4)') 'matinv_err ',maxval(abs(matmul(mat,inv)-reshape([1.d0,0.d0,0.d0,0.d0,1.d0,0.d0,0.d0,0.d0,1.d0],[3,3])))
 call svd(mat,u,sv,v)
 write(*,'(a,e12.4)') 'svd_err ',maxval(abs(matmul(u,matmul(reshape([sv(1),0.d0,0.d0,0.d0,sv(2),0.d0,0.d0,0.d0,sv(3)],[3,3]),v))-mat))
 write(*,'(a,i3)') 'chop ',chop([1.d0,1.d-3,1.d-9],tol=1.d-6)
 ! cos_approx_mod (lib/cos_approx.f90)
 phis=[(1.d0,0.d0),(0.5d0,-0.25d0),(-0.125d0,0.0625d0),(0.03d0,0.01d0)]
 call cos_approximate_array([0.5d0,1.25d0,2.75d0],phis,lower_bound=0.25d0,upper_bound=3.d0,n_terms=4,pdf_vals=pdf)
 write(*,'(a,3e24.16)') 'cos_array ',pdf
 write(*,'(a,e24.16)') 'cos_point ',cos_approximate(1.25d0,phis,0.25d0,3.d0,4)
 ! utils: save_dtt_to_hdf5 (lib/utils.f90) of a host train
 call get_command_argument(1,h5name)
 if(len_trim(h5name).gt.0)then
  call save_dtt_to_hdf5(a,trim(h5name))
  write(*,'(a)') 'hdf5_written'
 end if
 call dealloc(a); call dealloc(b); call dealloc(c)      ! each releases only its own storage
 write(*,'(a)') 'dealloc_ok'
end program
