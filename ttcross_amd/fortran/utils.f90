! utils.f90 -- drop-in for the reference's module utils (lib/utils.f90:8-57): save_dtt_to_hdf5(tt, filename) writes group
! "TT" with datasets "modes", "ranks" and "core_k".  The file is produced by the engine (ttx_write_hdf5, libhdf5 resolved
! at run time), from the device-resident train or from the host cores of a train that has no device handle.
module utils
 use iso_c_binding
 use tt_lib
 implicit none
 interface
  function ttx_write_hdf5(h,path) bind(C,name='ttx_write_hdf5') result(rc)
   import; type(c_ptr),value :: h; character(kind=c_char) :: path(*); integer(c_int) :: rc
  end function
 end interface
contains
 subroutine save_dtt_to_hdf5(tt,filename)
  use ttx_c, only: ttx_check,ttx_destroy
  type(dtt),intent(in) :: tt
  character(len=*),intent(in) :: filename
  character(kind=c_char) :: cnam(len_trim(filename)+1)
  type(c_ptr) :: h
  logical :: temp
  integer :: i
  do i=1,len_trim(filename); cnam(i)=filename(i:i); end do
  cnam(len_trim(filename)+1)=c_null_char
  call dtt_stage(tt,h,temp,'save_dtt_to_hdf5')
  call ttx_check(ttx_write_hdf5(h,cnam),'save_dtt_to_hdf5')
  if(temp)call ttx_destroy(h)
 end subroutine
end module
