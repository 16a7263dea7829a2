"""Probe of the (removed) diffuseness constraint from the reference's surviving fixtures: *_wDC vs *_woDC.

Per bin, W_dc(k) [C x 2 ears] = W_wo(k) M(k) with a 2x2 ear-mixing M(k).  This script fits M(k) from the fixture pairs,
derives the implied target covariance R(k) = M^H Rhat M for candidate definitions of the rendered covariance Rhat, checks
that R(k) agrees across the three methods (they share the HRIR set, so they must share R), and compares candidate closed
forms of M with the fit.  Output feeds oracle.diffuseness_mixing() and tests/test_oracle_kats.py."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import emagls_oracle as O   # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "ref_fixtures.npz"))
NFFT = 1024
P = NFFT // 2 + 1


def spec(name):
    w = g[name]
    return np.fft.fft(np.vstack([w, np.zeros((NFFT - w.shape[0], w.shape[1]))]), axis=0)[:P]


def fit_M(tag_wo, tag_dc, nl, nr):
    Wl, Wr = spec(f"{tag_wo}/{nl}"), spec(f"{tag_wo}/{nr}")
    Dl, Dr = spec(f"{tag_dc}/{nl}"), spec(f"{tag_dc}/{nr}")
    M = np.zeros((P, 2, 2), complex)
    res = np.zeros(P)
    for k in range(P):
        A = np.column_stack([Wl[k], Wr[k]])       # C x 2
        B = np.column_stack([Dl[k], Dr[k]])
        M[k] = np.linalg.lstsq(A, B, rcond=None)[0]
        res[k] = np.linalg.norm(A @ M[k] - B) / np.linalg.norm(B)
    return M, res, (Wl, Wr)


pairs = {
    "MagLS": ("real_MagLS_woDC", "real_MagLS_wDC", "wMlsL", "wMlsR"),
    "eMagLS": ("real_eMagLS_woDC", "real_eMagLS_wDC", "wEMlsL", "wEMlsR"),
    "eMagLS2": ("real_eMagLS2_woDC", "real_eMagLS2_wDC", "wEMls2L", "wEMls2R"),
}
fits = {m: fit_M(*p) for m, p in pairs.items()}
ks = [2, 5, 10, 20, 40, 43, 60, 100, 200, 300, 400, 500]
for m, (M, res, _) in fits.items():
    print(f"== {m}: residual median {np.median(res[1:-1]):.2e} max {res[1:-1].max():.2e}")
    for k in ks:
        herm = np.abs(M[k] - M[k].conj().T).max()
        print(f"  k={k:3d} M=[[{M[k,0,0]:.4f} {M[k,0,1]:.4f}] [{M[k,1,0]:.4f} {M[k,1,1]:.4f}]] |M-M^H|={herm:.1e} res={res[k]:.1e}")


# ---------------------------------------------------------------------------------------------
# rendered covariance Rhat(k) per method (in the final-filter domain) and the implied target R(k) = M^H Rhat M
# ---------------------------------------------------------------------------------------------
azi, zen = g["grid/hrirGridAziRad"], g["grid/hrirGridZenRad"]
maz, mzn = g["grid/micGridAziRad"], g["grid/micGridZenRad"]
D = azi.size
Yh = O.getSH(4, np.column_stack([azi, zen]), "real")            # D x 25
micgrid = np.column_stack([maz, mzn])
sm, simOrder = O.getSMAIRMatrix(4, 48000.0, NFFT, 0.042, micgrid, "real", False)
sm2, _ = O.getSMAIRMatrix(4, 48000.0, NFFT, 0.042, micgrid, "real", True)
Ysim_conj = O.getSH(simOrder, np.column_stack([azi, zen]), "real").conj().T      # S x D


def rendered(method, k):
    Wl, Wr = fits[method][2]
    if method == "MagLS":
        G = Yh.conj().T                                         # 25 x D
    elif method == "eMagLS":
        G = sm[:, :, k] @ Ysim_conj
    else:
        G = sm2[:, :, k] @ Ysim_conj
    return np.column_stack([Wl[k] @ G, Wr[k] @ G])              # D x 2 rendered HRTFs


def sqrtm_h(A):
    w, V = np.linalg.eigh(A)
    return (V * np.sqrt(np.maximum(w, 0))) @ V.conj().T


def hpd_solution(Rhat, R):
    """the Hermitian positive definite M with M Rhat M = R"""
    s = sqrtm_h(Rhat)
    si = np.linalg.inv(s)
    return si @ sqrtm_h(s @ R @ s) @ si


print("\n== implied target covariance R(k) = M^H Rhat M, Rhat = Hhat^H Hhat / D, per method (should agree: same HRIRs)")
Rest = {m: np.zeros((P, 2, 2), complex) for m in fits}
Rhat = {m: np.zeros((P, 2, 2), complex) for m in fits}
for m in fits:
    for k in range(1, P):
        Hh = rendered(m, k)
        Rhat[m][k] = Hh.conj().T @ Hh / D
        Mk = fits[m][0][k]
        Rest[m][k] = Mk.conj().T @ Rhat[m][k] @ Mk
for k in ks:
    a, b, c = Rest["MagLS"][k], Rest["eMagLS"][k], Rest["eMagLS2"][k]
    print(f"  k={k:3d} R11 {a[0,0].real:.5f} {b[0,0].real:.5f} {c[0,0].real:.5f} | R22 {a[1,1].real:.5f} {b[1,1].real:.5f} {c[1,1].real:.5f}"
          f" | R12 {a[0,1]:.4f} {b[0,1]:.4f} {c[0,1]:.4f}")
dev = lambda x, y: np.array([np.linalg.norm(x[k] - y[k]) / np.linalg.norm(x[k]) for k in range(2, P - 1)])
print("  rel. deviation MagLS vs eMagLS : median %.2e max %.2e" % (np.median(dev(Rest["MagLS"], Rest["eMagLS"])), dev(Rest["MagLS"], Rest["eMagLS"]).max()))
print("  rel. deviation eMagLS vs eMagLS2: median %.2e max %.2e" % (np.median(dev(Rest["eMagLS"], Rest["eMagLS2"])), dev(Rest["eMagLS"], Rest["eMagLS2"]).max()))

print("\n== closed form: M = HPD solution of M Rhat M = R, with R taken from ANOTHER method's fit (cross prediction)")
for m, src in (("MagLS", "eMagLS"), ("eMagLS", "MagLS"), ("eMagLS2", "eMagLS")):
    err = []
    for k in range(2, P - 1):
        Mp = hpd_solution(Rhat[m][k], Rest[src][k])
        err.append(np.linalg.norm(Mp - fits[m][0][k]) / np.linalg.norm(fits[m][0][k]))
    err = np.array(err)
    print(f"  {m} predicted with R from {src}: rel err median {np.median(err):.2e}  90% {np.quantile(err, .9):.2e}  max {err.max():.2e}")


print("\n== MagLS: Rhat in the SH domain, W W^H / (4 pi) (orthonormal SHs), instead of through the 2702-point grid")
Wl, Wr = fits["MagLS"][2]
R_sh = np.zeros((P, 2, 2), complex)
Rest_sh = np.zeros((P, 2, 2), complex)
for k in range(1, P):
    A = np.column_stack([Wl[k], Wr[k]])            # 25 x 2, columns = ears;  Hhat(d) = A^T conj(y_d)
    R_sh[k] = (A.T @ A.conj()).conj() / (4 * np.pi)    # E_d[conj(hhat_i) hhat_j] with E[conj(y) y^T] = I / 4pi
    Mk = fits["MagLS"][0][k]
    Rest_sh[k] = Mk.conj().T @ R_sh[k] @ Mk
print("  grid vs SH-domain Rhat (MagLS): median rel dev %.2e" % np.median(dev(Rhat["MagLS"], R_sh)))
print("  implied R, SH-domain MagLS vs eMagLS: median %.2e max %.2e" % (np.median(dev(Rest_sh, Rest["eMagLS"])), dev(Rest_sh, Rest["eMagLS"]).max()))
for k in ks:
    print(f"  k={k:3d} R12 grid {Rest['MagLS'][k][0,1]:.4f}  sh {Rest_sh[k][0,1]:.4f}  eMagLS {Rest['eMagLS'][k][0,1]:.4f}")


print("\n== conjugation conventions: implied R from (M, Rhat), (M, conj Rhat); MagLS vs eMagLS vs eMagLS2, medians over bins 2..P-2")
def implied(m, conj_rhat, conj_m):
    out = np.zeros((P, 2, 2), complex)
    for k in range(1, P):
        Mk = fits[m][0][k]
        Mk = Mk.conj() if conj_m else Mk
        Rh = Rhat[m][k].conj() if conj_rhat else Rhat[m][k]
        out[k] = Mk.conj().T @ Rh @ Mk
    return out
for ca, cb in itertools.product((False, True), repeat=2):
    a = implied("MagLS", ca, False)
    b = implied("eMagLS", cb, False)
    c = implied("eMagLS2", cb, False)
    print(f"  conj(Rhat): MagLS {ca!s:5} eMagLS {cb!s:5} -> MagLS vs eMagLS median {np.median(dev(a, b)):.2e} | vs conj {np.median(dev(a, b.conj())):.2e}"
          f" | eMagLS vs eMagLS2 {np.median(dev(b, c)):.2e}")
print("  Rhat12 at some bins (MagLS | eMagLS | eMagLS2):")
for k in ks:
    print(f"  k={k:3d} {Rhat['MagLS'][k][0,1]:.4f} | {Rhat['eMagLS'][k][0,1]:.4f} | {Rhat['eMagLS2'][k][0,1]:.4f}   diag {Rhat['MagLS'][k][0,0].real:.4f} {Rhat['eMagLS'][k][0,0].real:.4f} {Rhat['eMagLS2'][k][0,0].real:.4f}")


print("\n== MagLS: the Rhat' that would make M = HPD(Rhat', R) with the (physical, nearly real R12) R implied by eMagLS")
for k in ks:
    Mi = np.linalg.inv(fits["MagLS"][0][k])
    Rp = Mi.conj().T @ Rest["eMagLS"][k] @ Mi
    Rh = Rhat["MagLS"][k]
    print(f"  k={k:3d} needed [{Rp[0,0].real:.4f} {Rp[1,1].real:.4f} {Rp[0,1]:.4f}]  grid-rendered [{Rh[0,0].real:.4f} {Rh[1,1].real:.4f} {Rh[0,1]:.4f}]")


print("\n== edge bins (0-based): DC, first solved bin, last bins")
for m in fits:
    for k in (0, 1, 2, 510, 511, 512):
        M = fits[m][0][k]
        print(f"  {m:8s} k={k:3d} M=[[{M[0,0]:.4f} {M[0,1]:.4f}] [{M[1,0]:.4f} {M[1,1]:.4f}]] res={fits[m][1][k]:.1e}")


print("\n== MagLS: candidates for the cross term of Rhat in the SH domain (needed = M^-1 R M^-1 with R implied by eMagLS)")
Wl, Wr = fits["MagLS"][2]
for k in (100, 200, 300, 400):
    Mi = np.linalg.inv(fits["MagLS"][0][k])
    need = (Mi.conj().T @ Rest["eMagLS"][k] @ Mi)[0, 1]
    cands = {"conj(Wl).Wr": np.vdot(Wl[k], Wr[k]), "Wl.conj(Wr)": np.vdot(Wr[k], Wl[k]), "Wl.Wr": np.dot(Wl[k], Wr[k]),
             "conj(Wl).conj(Wr)": np.conj(np.dot(Wl[k], Wr[k]))}
    print(f"  k={k}: needed {need:.4f} |" + " | ".join(f"{n} {v / (4 * np.pi):.4f}" for n, v in cands.items()))


# ---------------------------------------------------------------------------------------------
# Round 5: the paper's closed form with Cholesky factors and an SVD (Zaunschirm, Schoerkhuber, Hoeldrich 2018,
# covariance constraint):  Rhat = Xh^H Xh,  R = X^H X (upper Cholesky factors),  M = Xh^-1 Q X  with the unitary Q = V U^H taken from
# U S V^H = svd(X^H Xh) (the choice that keeps Hhat M closest to Hhat).  Any unitary Q satisfies M^H Rhat M = R; this M is NOT
# Hermitian in general -- the MagLS fit is Hermitian only to 5 % -- so it is the one candidate the HPD solution above cannot cover.
# All four orders of the SVD argument and both Q = V U^H / U V^H are scored; R comes from another method's fit as above.
# ---------------------------------------------------------------------------------------------
print("\n== closed form M = Xh^-1 Q X (Cholesky factors + SVD), R from another method's fit; median / 90 % relative error of M over bins 44..500")


def chol_upper(A):
    A = 0.5 * (A + A.conj().T)
    return np.linalg.cholesky(A).conj().T          # A = X^H X


def chol_svd_solution(Rh, R, arg, q_form):
    Xh, X = chol_upper(Rh), chol_upper(R)
    T = {"XhXH": X.conj().T @ Xh, "XhHX": Xh.conj().T @ X, "XXhH": X @ Xh.conj().T, "XhXHr": Xh @ X.conj().T}[arg]
    U, _, Vh = np.linalg.svd(T)
    Q = (Vh.conj().T @ U.conj().T) if q_form == "VUH" else (U @ Vh)
    return np.linalg.inv(Xh) @ Q @ X


best = {}
for m, src in (("MagLS", "eMagLS"), ("eMagLS", "eMagLS2"), ("eMagLS2", "eMagLS")):
    for arg in ("XhXH", "XhHX", "XXhH", "XhXHr"):
        for q_form in ("VUH", "UVH"):
            err, ok = [], []
            for k in range(44, 501):
                try:
                    Mp = chol_svd_solution(Rhat[m][k], Rest[src][k], arg, q_form)
                except np.linalg.LinAlgError:
                    continue
                Mk = fits[m][0][k]
                err.append(np.linalg.norm(Mp - Mk) / np.linalg.norm(Mk))
                ok.append(np.linalg.norm(Mp.conj().T @ Rhat[m][k] @ Mp - Rest[src][k]) / np.linalg.norm(Rest[src][k]))
            err = np.array(err)
            best.setdefault(m, []).append((np.median(err), arg, q_form))
            print(f"  {m:8s} R from {src:8s} svd({arg:6s}) Q={q_form}: median {np.median(err):.2e}  90% {np.quantile(err, .9):.2e}  (constraint residual {np.median(ok):.1e})")
    hp = np.array([np.linalg.norm(hpd_solution(Rhat[m][k], Rest[src][k]) - fits[m][0][k]) / np.linalg.norm(fits[m][0][k]) for k in range(44, 501)])
    print(f"  {m:8s} R from {src:8s} HPD solution (what ships): median {np.median(hp):.2e}  90% {np.quantile(hp, .9):.2e}")
for m, lst in best.items():
    lst.sort()
    print(f"  best Cholesky + SVD form for {m}: svd({lst[0][1]}) Q={lst[0][2]}: median {lst[0][0]:.2e}")
