! mat_lib.f90 -- drop-in for the names of the reference's mat_lib (lib/mat.f90) that a driver can reach:
! matinv, svd (A = U S V with truncation), chop, eye, laplace, d2submat, d2subset, norm2 -- real double precision.
! The reference wraps LAPACK (dgesvd, dgetrf/dgetri); nothing here is on the device path (the three drivers of the
! scope `use mat_lib` without calling it), so these are small self-contained host routines: Gauss-Jordan with
! partial pivoting for the inverse and a one-sided Jacobi method for the SVD.
module mat_lib
 implicit none
 interface matinv;  module procedure matinv_d;  end interface
 interface svd;     module procedure d_svd;     end interface
 interface eye;     module procedure eye_d2,eye_z2; end interface
 interface laplace; module procedure laplace_d2; end interface
 private :: matinv_d,jacobi_svd
contains
 subroutine matinv_d(a,ainv,alg,tol)
  ! lib/mat.f90:23-44: alg='t' (or absent for a square matrix here) -> LU-type inverse; alg='s' or a rectangular
  ! matrix -> pseudo-inverse through the SVD, singular values below tol*s(1) dropped.  ainv absent: in place.
  double precision,intent(inout) :: a(:,:)
  double precision,intent(out),optional :: ainv(size(a,2),size(a,1))
  character(len=1),intent(in),optional :: alg
  double precision,intent(in),optional :: tol
  double precision,allocatable :: w(:,:),x(:,:)
  double precision,pointer :: u(:,:),s(:),v(:,:)
  double precision :: t,big,cut
  integer :: n,i,j,k,piv,r
  logical :: usesvd
  usesvd=size(a,1).ne.size(a,2)
  if(present(alg))then
   select case(alg)
    case('s','S'); usesvd=.true.
    case('t','T'); if(size(a,1).ne.size(a,2))then;write(*,*)'matinv_d: alg t needs a square matrix';stop;endif
    case default; write(*,'(3a)')'matinv_d',': unknown alg: ',alg; stop
   end select
  else
   usesvd=.true.                                   ! the reference's default (lib/mat.f90:31-32)
  end if
  if(usesvd)then
   call d_svd(a,u,s,v)
   cut=1.d-14; if(present(tol))cut=tol
   r=count(s.gt.cut*s(1))
   allocate(x(size(a,2),size(a,1)))
   x=0.d0
   do k=1,r
    do j=1,size(a,1); do i=1,size(a,2); x(i,j)=x(i,j)+v(k,i)*u(j,k)/s(k); end do; end do
   end do
   deallocate(u,s,v)
  else
   n=size(a,1)
   allocate(w(n,2*n),x(n,n))
   w(:,1:n)=a; w(:,n+1:2*n)=0.d0
   do i=1,n; w(i,n+i)=1.d0; end do
   do k=1,n
    piv=k; big=abs(w(k,k))
    do i=k+1,n; if(abs(w(i,k)).gt.big)then; big=abs(w(i,k)); piv=i; endif; end do
    if(big.eq.0.d0)then;write(*,*)'matinv_d: singular matrix';stop;endif
    if(piv.ne.k)then; do j=1,2*n; t=w(k,j); w(k,j)=w(piv,j); w(piv,j)=t; end do; endif
    t=1.d0/w(k,k); w(k,:)=w(k,:)*t
    do i=1,n
     if(i.ne.k)then; t=w(i,k); if(t.ne.0.d0)w(i,:)=w(i,:)-t*w(k,:); endif
    end do
   end do
   x=w(:,n+1:2*n)
  end if
  if(present(ainv))then; ainv=x
  else
   if(size(a,1).ne.size(a,2))then;write(*,*)'matinv_d: in-place inverse of a rectangular matrix';stop;endif
   a=x
  end if
 end subroutine

 subroutine jacobi_svd(a,uu,ss,vv)
  ! one-sided Jacobi on the columns of a (m x n, m >= n): a = uu diag(ss) vv^T, ss descending
  double precision,intent(in) :: a(:,:)
  double precision,intent(out) :: uu(size(a,1),size(a,2)),ss(size(a,2)),vv(size(a,2),size(a,2))
  double precision :: al,be,ga,ze,t,c,s,x,y
  integer :: m,n,i,p,q,sweep,k
  logical :: rotated
  m=size(a,1); n=size(a,2)
  uu=a; vv=0.d0
  do i=1,n; vv(i,i)=1.d0; end do
  do sweep=1,60
   rotated=.false.
   do p=1,n-1
    do q=p+1,n
     al=dot_product(uu(:,p),uu(:,p)); be=dot_product(uu(:,q),uu(:,q)); ga=dot_product(uu(:,p),uu(:,q))
     if(abs(ga).le.1.d-15*sqrt(al*be).or.ga.eq.0.d0)cycle
     rotated=.true.
     ze=(be-al)/(2.d0*ga); t=sign(1.d0,ze)/(abs(ze)+sqrt(1.d0+ze*ze)); c=1.d0/sqrt(1.d0+t*t); s=c*t
     do i=1,m; x=uu(i,p); y=uu(i,q); uu(i,p)=c*x-s*y; uu(i,q)=s*x+c*y; end do
     do i=1,n; x=vv(i,p); y=vv(i,q); vv(i,p)=c*x-s*y; vv(i,q)=s*x+c*y; end do
    end do
   end do
   if(.not.rotated)exit
  end do
  do p=1,n; ss(p)=sqrt(dot_product(uu(:,p),uu(:,p))); end do
  do p=1,n-1                                        ! selection sort, descending
   k=p
   do q=p+1,n; if(ss(q).gt.ss(k))k=q; end do
   if(k.ne.p)then
    t=ss(p); ss(p)=ss(k); ss(k)=t
    do i=1,m; t=uu(i,p); uu(i,p)=uu(i,k); uu(i,k)=t; end do
    do i=1,n; t=vv(i,p); vv(i,p)=vv(i,k); vv(i,k)=t; end do
   end if
  end do
  do p=1,n; if(ss(p).gt.0.d0)uu(:,p)=uu(:,p)/ss(p); end do
 end subroutine

 subroutine d_svd(a,u,s,v,tol,rmax,err,info)
  ! lib/mat.f90:340-385: a(m,n) = u(m,r) diag(s(r)) v(r,n), r chosen by chop(s,tol,rmax); u,s,v are allocated here
  double precision,intent(in) :: a(:,:)
  double precision,pointer :: u(:,:),s(:),v(:,:)
  double precision,intent(in),optional :: tol
  integer,intent(in),optional :: rmax
  double precision,intent(out),optional :: err
  integer,intent(out),optional :: info
  double precision,allocatable :: uu(:,:),ss(:),vv(:,:)
  integer :: m,n,mn,r,i
  m=size(a,1); n=size(a,2); mn=min(m,n)
  if(present(info))info=0
  if(m.ge.n)then
   allocate(uu(m,n),ss(n),vv(n,n)); call jacobi_svd(a,uu,ss,vv)
  else
   allocate(uu(n,m),ss(m),vv(m,m)); call jacobi_svd(transpose(a),uu,ss,vv)   ! a^T = uu ss vv^T  =>  a = vv ss uu^T
  end if
  r=chop(ss(1:mn),tol,rmax,err)
  allocate(u(m,r),s(r),v(r,n))
  s=ss(1:r)
  if(m.ge.n)then
   u=uu(:,1:r); do i=1,r; v(i,:)=vv(:,i); end do
  else
   u=vv(:,1:r); do i=1,r; v(i,:)=uu(:,i); end do
  end if
 end subroutine

 integer function chop(s,tol,rmax,err) result(r)
  ! lib/mat.f90:433-458: smallest rank whose discarded tail obeys  sum s(k)^2 < tol^2 |s|^2  (and r <= rmax)
  double precision,intent(in) :: s(:)
  double precision,intent(in),optional :: tol
  integer,intent(in),optional :: rmax
  double precision,intent(out),optional :: err
  double precision :: total,tail,nexttail
  r=size(s); tail=0.d0
  if(present(rmax))then
   if(rmax.lt.r)then; tail=sum(s(rmax+1:r)**2); r=rmax; endif
  end if
  if(present(tol))then
   total=tol*tol*sum(s**2)
   do while(r.ge.1)
    nexttail=tail+s(r)*s(r)
    if(nexttail.ge.total)exit
    tail=nexttail; r=r-1
   end do
  end if
  if(present(err))err=sqrt(tail)
 end function

 subroutine eye_d2(a)
  double precision,intent(inout) :: a(:,:)
  integer :: i
  a=0.d0
  do i=1,min(size(a,1),size(a,2)); a(i,i)=1.d0; end do
 end subroutine
 subroutine eye_z2(a)
  double complex,intent(inout) :: a(:,:)
  integer :: i
  a=(0.d0,0.d0)
  do i=1,min(size(a,1),size(a,2)); a(i,i)=(1.d0,0.d0); end do
 end subroutine
 subroutine d2eye(a,n)
  integer,intent(in) :: n
  double precision,intent(out) :: a(n,n)
  call eye_d2(a)
 end subroutine
 subroutine laplace_d2(a)
  ! tridiagonal (-1, 2, -1)
  double precision,intent(out) :: a(:,:)
  integer :: i,m,n
  m=size(a,1); n=size(a,2); a=0.d0
  do i=1,min(m,n); a(i,i)=2.d0; end do
  do i=1,min(m-1,n); a(i+1,i)=-1.d0; end do
  do i=1,min(m,n-1); a(i,i+1)=-1.d0; end do
 end subroutine
 subroutine d2submat(m,n,a,lda,b)
  integer,intent(in) :: m,n,lda
  double precision,intent(in) :: a(lda,n)
  double precision,intent(out) :: b(m,n)
  if(m.gt.lda)then;write(*,*)'d2submat: lda,m: ',lda,m;stop;endif
  b(1:m,1:n)=a(1:m,1:n)
 end subroutine
 subroutine d2subset(m,n,r,ind,jnd,a,b)
  integer,intent(in) :: m,n,r,ind(r),jnd(r)
  double precision,intent(in) :: a(m,n)
  double precision,intent(out) :: b(r,r)
  integer :: i,j
  do j=1,r; do i=1,r; b(i,j)=a(ind(i),jnd(j)); end do; end do
 end subroutine
 double precision function norm2_d(a) result(nrm)
  ! spectral norm = largest singular value
  double precision,intent(in) :: a(:,:)
  double precision,pointer :: u(:,:),s(:),v(:,:)
  call d_svd(a,u,s,v)
  nrm=s(1)
  deallocate(u,s,v)
 end function
end module
