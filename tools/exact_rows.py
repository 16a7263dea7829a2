"""Least-squares rows of the eMagLS per-bin solve in 40-digit arithmetic (dev / test tooling, CPU only; mpmath).

The reference forms pwGrid = smairMat(:,:,k) * Y_Hi_conj in FP64 (a BLAS product), takes LAPACK's SVD of the rounded matrix, clips
the singular values at 1 % of the largest and multiplies H(k,:) into conj(U) (s .* V.') (lib/getEMagLs2Filters.m:85-99).  Where the
array model spans 12+ decades (k r << 1 with more microphones than low-order SH channels) the singular vectors of the small singular
values -- which the clipping weights with 100 / s_max -- are decided by the rounding of that product, i.e. the reference's own result
is defined only to some accuracy.  `exact_ls_rows` takes the SAME double-precision factors as exact numbers and carries the product,
the SVD and the row through 40 digits: the distance between the FP64 oracle and this row measures that accuracy, and an
implementation closer to the exact row than the oracle is cannot be told apart from the reference by the reference's own arithmetic."""
import numpy as np


def _to_mp(a):
    import mpmath as mp
    a = np.atleast_2d(np.asarray(a))
    return mp.matrix([[mp.mpc(float(np.real(v)), float(np.imag(v))) for v in row] for row in a])


def exact_ls_rows(pM, Yc, H_rows, regul=0.01, dps=40):
    """pM: smairMat(:,:,k) (C x S), Yc: Y_Hi_conj (S x D), H_rows: iterable of H(k,:) (D,) -- all FP64 values taken as exact.
    Returns ([H Y_reg_inv for H in H_rows] as complex128 arrays (C,), the exact singular values of pwGrid.' as float64)."""
    import mpmath as mp
    with mp.workdps(dps):
        C = pM.shape[0]
        G = (_to_mp(pM) * _to_mp(Yc)).T          # D x C = pwGrid.'
        Q, R = mp.qr(G, mode="skinny")           # D x C, C x C
        U2, s, Vh = mp.svd_c(R)                  # R = U2 diag(s) Vh;  U = Q U2, V.' = conj(Vh)
        smax = max(s)
        sreg = [1 / max(x, mp.mpf(regul) * smax) for x in s]
        Qc, U2c, Vhc = Q.apply(mp.conj), U2.apply(mp.conj), Vh.apply(mp.conj)
        rows = []
        for H in H_rows:
            t = (_to_mp(np.asarray(H)[None, :]) * Qc) * U2c     # H conj(U)
            for j in range(C):
                t[0, j] = t[0, j] * sreg[j]
            w = t * Vhc
            rows.append(np.array([complex(w[0, j]) for j in range(C)]))
        return rows, np.array([float(x) for x in s])


def oracle_ls_rows(pM, Yc, H_rows, regul=0.01):
    """The same rows the way the oracle (and the reference) computes them: FP64 product, LAPACK SVD of the rounded matrix."""
    A = np.ascontiguousarray(pM)
    pw = (A.real @ Yc) + 1j * (A.imag @ Yc) if np.isrealobj(Yc) else A @ Yc
    U, s, Vh = np.linalg.svd(pw.T, full_matrices=False)
    yri = np.conj(U) @ ((1.0 / np.maximum(s, regul * s.max()))[:, None] * Vh.conj())
    return [np.asarray(H) @ yri for H in H_rows], s
