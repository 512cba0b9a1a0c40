"""numpy restatement of Edward Snelson's SPGP likelihood, the third-party Octave routine the reference keeps in
its test directory (test/spgp_lik.m, helper test/dist.m) and compares its own numbers with in test/oct.m:183-191.

Test infrastructure only: an implementation of the same FITC/SPGP evidence and gradient that shares no formula
with the reference's QR-based sequence (explicit inverses, a different parameterisation: one inverse squared
length scale b_d per dimension, amplitude c, noise sig, jitter del on the pseudo-input covariance).  Agreement of
the oracle -- and through it the HIP path -- with this routine is the one external pin the reference's own tests
offer.  Variable names follow spgp_lik.m; comments give its line numbers.
"""
import numpy as np


def _dist(x0, x1):
    """test/dist.m: D[i, j] = x0[i] - x1[j]."""
    return x0[:, None] - x1[None, :]


def spgp_lik(w, y, x, n, jitter=1e-6):
    """spgp_lik.m:3-122.  x: N x dim, w = [xb (n x dim, column-major); log b (dim); log c; log sig].
    Returns (fw, dfw) = negative log likelihood and its gradient in the layout of w."""
    y = np.asarray(y, dtype=np.float64).copy()
    N, dim = x.shape
    xb = np.reshape(w[:n * dim], (n, dim), order="F")                     # :32
    b = np.exp(w[n * dim:n * dim + dim])
    c = np.exp(w[-2])
    sig = np.exp(w[-1])                                                   # :33
    xb = xb * np.sqrt(b)[None, :]                                         # :35
    x = x * np.sqrt(b)[None, :]                                           # :36
    Q = xb @ xb.T                                                         # :38-40
    Q = np.diag(Q)[:, None] + np.diag(Q)[None, :] - 2 * Q
    Q = c * np.exp(-0.5 * Q) + jitter * np.eye(n)
    K = -2 * xb @ x.T + np.sum(x * x, axis=1)[None, :] + np.sum(xb * xb, axis=1)[:, None]   # :42-43
    K = c * np.exp(-0.5 * K)
    L = np.linalg.cholesky(Q)                                             # :45  chol(Q)'
    V = np.linalg.solve(L, K)                                             # :46
    ep = 1 + (c - np.sum(V ** 2, axis=0)) / sig                           # :47
    K = K / np.sqrt(ep)[None, :]                                          # :48
    V = V / np.sqrt(ep)[None, :]                                          # :49
    y = y / np.sqrt(ep)
    Lm = np.linalg.cholesky(sig * np.eye(n) + V @ V.T)                    # :50
    invLmV = np.linalg.solve(Lm, V)                                       # :51
    bet = invLmV @ y                                                      # :52
    fw = (np.sum(np.log(np.diag(Lm))) + (N - n) / 2 * np.log(sig) + (y @ y - bet @ bet) / 2 / sig
          + np.sum(np.log(ep)) / 2 + 0.5 * N * np.log(2 * np.pi))         # :55-56
    # ---- derivatives, :61-120
    Lt = L @ Lm                                                           # :62
    B1 = np.linalg.solve(Lt.T, invLmV)                                    # :63
    b1 = np.linalg.solve(Lt.T, bet)                                       # :64
    invLV = np.linalg.solve(L.T, V)                                       # :65
    invL = np.linalg.inv(L)
    invQ = invL.T @ invL                                                  # :66
    invLt = np.linalg.inv(Lt)
    invA = invLt.T @ invLt                                                # :67
    mu = (np.linalg.solve(Lm.T, bet) @ V)                                 # :68
    sumVsq = np.sum(V ** 2, axis=0)                                       # :69
    bigsum = (y * (bet @ invLmV) / sig - np.sum(invLmV * invLmV, axis=0) / 2 - (y ** 2 + mu ** 2) / 2 / sig
              + 0.5)                                                      # :70-71
    TT = invLV @ (invLV.T * bigsum[:, None])                              # :72
    dfxb = np.zeros((n, dim))
    dfb = np.zeros(dim)
    for i in range(dim):                                                  # :75-100
        dnnQ = _dist(xb[:, i], xb[:, i]) * Q                              # :78
        dNnK = _dist(-xb[:, i], -x[:, i]) * K                             # :79
        epdot = -2 / sig * dNnK * invLV                                   # :81
        epPmod = -np.sum(epdot, axis=0)
        dfxb[:, i] = (-b1 * (dNnK @ (y - mu) / sig + dnnQ @ b1) + np.sum((invQ - invA * sig) * dnnQ, axis=1)
                      + epdot @ bigsum - 2 / sig * np.sum(dnnQ * TT, axis=1))         # :83-85
        dfb[i] = (((y - mu) * (b1 @ dNnK)) / sig + epPmod * bigsum) @ x[:, i]         # :87-88
        dNnK = dNnK * B1                                                  # :90
        dfxb[:, i] += np.sum(dNnK, axis=1)                                # :91
        dfb[i] -= np.sum(dNnK, axis=0) @ x[:, i]                          # :92
        dfxb[:, i] *= np.sqrt(b[i])                                       # :94
        dfb[i] /= np.sqrt(b[i])                                           # :96
        dfb[i] += dfxb[:, i] @ xb[:, i] / b[i]                            # :97
        dfb[i] *= np.sqrt(b[i]) / 2                                       # :98
    epc = (c / ep - sumVsq - jitter * np.sum(invLV ** 2, axis=0)) / sig   # :103
    dfc = ((n + jitter * np.trace(invQ - sig * invA) - sig * np.sum(invA * Q.T)) / 2 - mu @ (y - mu) / sig
           + b1 @ (Q - jitter * np.eye(n)) @ b1 / 2 + epc @ bigsum)       # :105-108
    dfsig = np.sum(bigsum / ep)                                           # :111
    dfw = np.concatenate([dfxb.ravel(order="F"), dfb, [dfc], [dfsig]])    # :113
    return float(fw), dfw
