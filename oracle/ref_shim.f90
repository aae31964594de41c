! Test infrastructure only (oracle/_ref): bind(C) entry points that forward to the
! reference's own `module sympgpr` (python/05_tokamak/SympGPR/sympgpr.f90:12-60),
! whose procedures take assumed-shape arrays and therefore cannot be called from C
! directly.  This file is OUR code; the reference sources are compiled where they
! lie under /root/reference by oracle/Makefile and are never copied into the repo.
module sympgpr_ref_shim
    use iso_c_binding
    use sympgpr, only: build_K, buildKreg, guessP, calcq, calcP, applymap_tok
    use fieldlines, only: fl_init => init, fl_ath => Ath, fl_timestep => timestep, fl_compute_r => compute_r
    implicit none
contains

subroutine ref_build_k(n, n0, x, y, x0, y0, hyp, K) bind(C, name="ref_build_k")
    integer(c_int), value :: n, n0
    real(c_double), intent(in) :: x(n), y(n), x0(n0), y0(n0), hyp(3)
    real(c_double), intent(inout) :: K(2*n, 2*n0)
    call build_K(x, y, x0, y0, hyp, K)
end subroutine

subroutine ref_buildkreg(n, n0, x, y, x0, y0, hyp, K) bind(C, name="ref_buildkreg")
    integer(c_int), value :: n, n0
    real(c_double), intent(in) :: x(n), y(n), x0(n0), y0(n0), hyp(3)
    real(c_double), intent(inout) :: K(n, n0)
    call buildKreg(x, y, x0, y0, hyp, K)
end subroutine

function ref_guessp(x, y, hypp, np, xtrainp, ytrainp, ztrainp, Kyinvp) &
        bind(C, name="ref_guessp") result(r)
    integer(c_int), value :: np
    real(c_double), intent(in) :: x(1), y(1), hypp(3)
    real(c_double), intent(in) :: xtrainp(np), ytrainp(np), ztrainp(np), Kyinvp(np, np)
    real(c_double) :: r
    r = guessP(x, y, hypp, xtrainp, ytrainp, ztrainp, Kyinvp)
end function

function ref_calcq(x, y, nt, xtrain, ytrain, hyp, Kyinv, ztrain) &
        bind(C, name="ref_calcq") result(r)
    integer(c_int), value :: nt
    real(c_double), intent(in) :: x(1), y(1), hyp(3)
    real(c_double), intent(in) :: xtrain(nt), ytrain(nt), ztrain(2*nt), Kyinv(2*nt, 2*nt)
    real(c_double) :: r
    r = calcq(x, y, xtrain, ytrain, hyp, Kyinv, ztrain)
end function

function ref_calcp(x, y, hyp, hypp, np, xtrainp, ytrainp, ztrainp, Kyinvp, &
        nt, xtrain, ytrain, ztrain, Kyinv) bind(C, name="ref_calcp") result(r)
    integer(c_int), value :: np, nt
    real(c_double), intent(in) :: x(1), y(1), hyp(3), hypp(3)
    real(c_double), intent(in) :: xtrainp(np), ytrainp(np), ztrainp(np), Kyinvp(np, np)
    real(c_double), intent(in) :: xtrain(nt), ytrain(nt), ztrain(2*nt), Kyinv(2*nt, 2*nt)
    real(c_double) :: r
    r = calcP(x, y, hyp, hypp, xtrainp, ytrainp, ztrainp, Kyinvp, &
        xtrain, ytrain, ztrain, Kyinv)
end function

! applymap_tok (sympgpr.f90:128-177): qmap, pmap are [nm, Ntest, 1] in/out, as the f2py wrapper passes them
subroutine ref_applymap_tok(nm, ntest, hyp, hypp, Q0map, P0map, np, xtrainp, ytrainp, ztrainp, Kyinvp, &
        nt, xtrain, ytrain, ztrain, Kyinv, qmap, pmap) bind(C, name="ref_applymap_tok")
    integer(c_int), value :: nm, ntest, np, nt
    real(c_double), intent(in) :: hyp(3), hypp(3), Q0map(ntest), P0map(ntest)
    real(c_double), intent(in) :: xtrainp(np), ytrainp(np), ztrainp(np), Kyinvp(np, np)
    real(c_double), intent(in) :: xtrain(nt), ytrain(nt), ztrain(2*nt), Kyinv(2*nt, 2*nt)
    real(c_double), intent(inout) :: qmap(nm, ntest, 1), pmap(nm, ntest, 1)
    integer :: nm_, ntest_
    nm_ = nm
    ntest_ = ntest
    call applymap_tok(nm_, ntest_, hyp, hypp, Q0map, P0map, xtrainp, ytrainp, ztrainp, Kyinvp, &
        xtrain, ytrain, ztrain, Kyinv, qmap, pmap)
end subroutine

! module fieldlines (python/05_tokamak/Split_SympGPR/fieldlines.f90, the same file as SympGPR/fieldlines.f90): what
! calc_fieldlines.py:28-40 and Split_SympGPR/func.py:211 call through f2py
subroutine ref_fl_init(nph, am, an, aeps, aphase, arlast) bind(C, name="ref_fl_init")
    integer(c_int), value :: nph, am, an
    real(c_double), value :: aeps, aphase, arlast
    integer :: nph_, am_, an_
    real(c_double) :: aeps_, aphase_, arlast_
    nph_ = nph; am_ = am; an_ = an
    aeps_ = aeps; aphase_ = aphase; arlast_ = arlast
    call fl_init(nph_, am_, an_, aeps_, aphase_, arlast_)
end subroutine

function ref_fl_ath(r, th, ph) bind(C, name="ref_fl_ath") result(v)
    real(c_double), value :: r, th, ph
    real(c_double) :: v
    v = fl_ath(r, th, ph)
end function

subroutine ref_fl_timestep(z) bind(C, name="ref_fl_timestep")
    real(c_double), intent(inout) :: z(3)
    call fl_timestep(z)
end subroutine

function ref_compute_r(z, rstart) bind(C, name="ref_compute_r") result(r)
    real(c_double), intent(in) :: z(3)
    real(c_double), value :: rstart
    real(c_double) :: r
    r = fl_compute_r(z, rstart)
end function

end module sympgpr_ref_shim
