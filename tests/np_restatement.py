"""Second, independent restatement of the reference algorithm in NumPy matrix form (float64).

Used only to cross-check oracle/gpc_oracle.c (different loop structure, BLAS-backed products) and to
generate the small golden fixtures in tests/golden/ (tests/gen_golden.py).  Cites /root/reference files.
"""
import numpy as np

F = lambda s: float(np.float32(s))  # the reference's float literals promoted to double


def rbf(p0, p1, A, B):
    """src/rbf_kernel.cpp:15-18 for all pairs: A (a,2), B (b,2) -> (a,b)."""
    d = A[:, None, :] - B[None, :, :]
    sq = d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]
    return p0 * np.exp(-0.5 / p1 * sq)


# ---------------------------------------------------------------- dense GP (src/gaussian_process.cpp:15-45)

def dense_fit(X, y, sigmaf_sq=0.05 * 0.05, l_sq=9.0, sigman_sq=0.04 * 0.04, double_noise=True):
    n = X.shape[0]
    K = rbf(sigmaf_sq, l_sq, X, X)
    K[np.diag_indices(n)] += sigman_sq            # covariance_matrix(..., training=true)  :59-61
    if double_noise:
        K[np.diag_indices(n)] += sigman_sq        # C.diagonal() += sigman_sq              :21
    L = np.linalg.cholesky(K)
    z = np.linalg.solve(L, y.T)
    alpha = np.linalg.solve(L.T, z).T
    return L, alpha


def dense_predict(X, L, alpha, Xs, sigmaf_sq=0.05 * 0.05, l_sq=9.0):
    Ks = rbf(sigmaf_sq, l_sq, X, Xs)              # n x m
    f = alpha @ Ks                                # (ny, m)
    v = np.linalg.solve(L, Ks)
    V = sigmaf_sq - np.sum(v * v, axis=0)
    return f, V


# ---------------------------------------------------------------- sparse online GP (src/sparse_gp.hpp, sparse_gp_field.hpp)

class SparseGP:
    def __init__(self, ny=1, capacity=100, s20=None, eps_tol=None, p0=F(100.0), p1=1.0,
                 field_delete_bug=True, probit=False):
        self.ny, self.capacity, self.p0, self.p1 = ny, capacity, p0, p1
        self.s20 = (F(1e-1) if ny == 1 else F(1e2)) if s20 is None else s20
        self.eps_tol = (F(1e-6) if ny == 1 else F(1e-4)) if eps_tol is None else eps_tol
        self.field_delete_bug = field_delete_bug
        self.probit = probit
        self.reset()

    def reset(self):
        self.b = 0
        self.alpha = np.zeros((0, self.ny))
        self.C = np.zeros((0, 0))
        self.Q = np.zeros((0, 0))
        self.BV = np.zeros((0, 2))
        self.n_full = self.n_sparse = self.n_deleted = 0

    def _k(self, x):
        return rbf(self.p0, self.p1, self.BV, x[None, :])[:, 0]

    def _noise(self, y, m, s2):
        if self.probit:                                   # src/probit_noise.cpp:11-31
            import math
            with np.errstate(all="ignore"):
                sigma2 = np.float64(self.s20 + s2)
                sigma = np.sqrt(sigma2)
                z = np.float64(y[0] * m[0] / sigma)
                two_sqrt2 = float(np.float32(2.0) * np.sqrt(np.float32(2.0)))
                ef = np.float64(math.erf(z) if np.isfinite(z) else np.nan) / two_sqrt2
                efp = np.exp(-z * z / 2) / math.sqrt(2.0 * math.pi)
                q = np.array([y[0] / sigma * efp / ef])
                first = efp / ef
                r = ((-z * efp) / ef - first * first) / sigma2
            return float(r), q
        r = F(-1.0) / (self.s20 + s2)                     # src/gaussian_noise.cpp:15-18
        q = (y - m) / (self.s20 + s2)                     # :9-12
        return r, q

    def add(self, x, y):                                  # src/sparse_gp.hpp:89-249
        x = np.asarray(x, float)
        y = np.atleast_1d(np.asarray(y, float))
        kstar = self.p0
        if self.b == 0:
            self.alpha = (y / (kstar + self.s20))[None, :]
            self.C = np.array([[-1.0 / (kstar + self.s20)]])
            self.Q = np.array([[1.0 / kstar]])
            self.BV = x[None, :].copy()
            self.b = 1
            self.n_full += 1
            return
        k = self._k(x)
        m = self.alpha.T @ k
        s2 = kstar + k @ self.C @ k
        r, q = self._noise(y, m, s2)
        e_hat = self.Q @ k
        gamma = kstar - k @ e_hat
        if gamma < F(1e-12):
            gamma = 0.0
        if gamma < self.eps_tol and self.capacity != -1:
            eta = 1.0 / (1.0 + gamma * r)
            s_hat = self.C @ k + e_hat
            self.alpha = self.alpha + np.outer(s_hat, q * eta)
            self.C = self.C + r * eta * np.outer(s_hat, s_hat)
            self.n_sparse += 1
        else:
            s = np.concatenate([self.C @ k, [1.0]])
            self.alpha = np.vstack([self.alpha, np.zeros((1, self.ny))]) + np.outer(s, q)
            b = self.b
            Cn = np.zeros((b + 1, b + 1))
            Cn[:b, :b] = self.C
            self.C = Cn + r * np.outer(s, s)
            self.BV = np.vstack([self.BV, x[None, :]])
            Qn = np.zeros((b + 1, b + 1))
            Qn[:b, :b] = self.Q
            e = np.concatenate([e_hat, [-1.0]])
            with np.errstate(divide="ignore", invalid="ignore"):
                self.Q = Qn + (1.0 / gamma if gamma != 0 else np.inf) * np.outer(e, e)
            self.b += 1
            self.n_full += 1
        while self.b > self.capacity and self.capacity > 0:            # :206-223
            score = np.sum(self.alpha * self.alpha, axis=1) / (np.diag(self.Q) + np.diag(self.C))
            self.delete_bv(int(np.argmin(score)))                      # first minimum, like the strict '<'
        minscore = 0.0
        while minscore < F(1e-9) and self.b > 1:                        # :226-242
            score = 1.0 / np.diag(self.Q)
            loc = int(np.argmin(score))
            minscore = score[loc]
            if minscore < F(1e-9):
                self.delete_bv(loc)

    def delete_bv(self, loc):                             # src/sparse_gp.hpp:252-295
        b = self.b
        last = b - 1
        keep = list(range(b))
        # "swap loc to the last spot": position loc now holds what was last; the last index disappears
        keep[loc] = last
        keep = keep[:last]
        alphastar = self.alpha[loc].copy()
        cstar = self.C[loc, loc]
        qstar = self.Q[loc, loc]
        Cstar = self.C[keep, loc].copy()
        Qstar = self.Q[keep, loc].copy()
        if loc != last:
            # Cstar(loc) = Cstar(last): the element at position loc is C(last, loc)
            Cstar[loc] = self.C[last, loc]
            Qstar[loc] = self.Q[last, loc]
        C = self.C[np.ix_(keep, keep)].copy()
        Q = self.Q[np.ix_(keep, keep)].copy()
        alpha = self.alpha[keep].copy()
        if self.ny == 1 or not self.field_delete_bug:
            alpha = alpha - np.outer(Qstar + Cstar, alphastar) / (qstar + cstar)
        else:                                             # src/sparse_gp_field.hpp:250-253 (F8)
            alpha = alpha - np.outer((qstar + cstar) * (Qstar + Cstar), alphastar)
        C = C + np.outer(Qstar, Qstar) / qstar - np.outer(Qstar + Cstar, Qstar + Cstar) / (qstar + cstar)
        Q = Q - np.outer(Qstar, Qstar) / qstar
        self.alpha, self.C, self.Q = alpha, C, Q
        self.BV = self.BV[keep].copy()
        self.b = last
        self.n_deleted += 1

    def add_measurements(self, X, Y, perm=None):          # src/sparse_gp.hpp:59-86, explicit order (F7)
        Y = np.asarray(Y, float).reshape(X.shape[0], -1)
        order = range(X.shape[0]) if perm is None else perm
        for i in order:
            self.add(X[i], Y[i])

    def predict(self, Xs, conf=False):                    # src/sparse_gp.hpp:299-351
        m = Xs.shape[0]
        kstar = self.p0
        if self.b == 0:
            f = np.zeros((m, self.ny))
            sigma = np.full(m, kstar + self.s20)
        else:
            K = rbf(self.p0, self.p1, self.BV, Xs)        # b x m
            f = (self.alpha.T @ K).T
            sigma = self.s20 + kstar + np.einsum("im,ij,jm->m", K, self.C, K)
        sigma = np.where(sigma < 0, 0.0, sigma)
        if conf:
            sigma = F(100.0) * (1.0 - sigma / (kstar + self.s20))
        else:
            sigma = np.sqrt(sigma)
        return f, sigma


def sattolo_like(n, rs):
    """src/sparse_gp.hpp:43-56 driven by an explicit stream of rand() values."""
    ind = list(range(n))
    t = 0
    for i in range(n - 1, 0, -1):
        r = int(rs[t]) % i
        t += 1
        ind[i], ind[r] = ind[r], ind[i]
    return np.array(ind, dtype=np.int32)


def grid(res, sz):
    """src/gp_compressor.cpp:317-332."""
    xs = np.arange(sz, dtype=float)
    g = res * ((xs + 0.5) / sz - 0.5)
    X0 = np.tile(g, sz)          # x inner
    X1 = np.repeat(g, sz)        # y outer
    return X0, X1


def train_sigmaf_np(p0, p1, s20, alpha, Cm, BV, q0, q1, y, step, max_counter):
    """the live part of train_parameters (src/sparse_gp.hpp:586-640), NumPy, on a given state"""
    if len(BV) < 20:
        return p0, 0, np.zeros(max_counter + 2), np.zeros(2)
    d2 = (q0[:, None] - BV[None, :, 0]) ** 2 + (q1[:, None] - BV[None, :, 1]) ** 2
    e = np.exp(np.float64(np.float32(-0.5)) / p1 * d2)                      # n x b
    ls = np.zeros(max_counter + 2)
    c0 = 0.5 * np.log(2.0 * np.pi)
    counter = 0
    while True:
        ak = (p0 * e) @ alpha
        kd0 = e @ alpha
        kd1 = (p0 * 0.5 / (p1 * p1) * d2 * e) @ alpha
        delta = np.array([np.sum((ak - y) * kd0), np.sum((ak - y) * kd1)])
        p0 = p0 + step * delta[0]
        k = p0 * e
        sigma = s20 + p0 + np.einsum("ij,jk,ik->i", k, Cm, k)
        ls[counter] = np.sum(-c0 - 0.5 * np.log(sigma) - 0.5 * (y - k @ alpha) ** 2 / sigma)
        it = counter + 1
        if counter > max_counter:
            break
        counter += 1
        if not np.sqrt(delta @ delta) > np.float64(np.float32(1e-2)):
            break
    return p0, it, ls, delta


# ---------------------------------------------------------------- C5: Laplace mode of the probit GP, Rasmussen & Williams Alg. 3.1

def probit_functor(y, f, s20, std_phi):
    """q = dx_ln, r = dx2_ln of src/probit_noise.cpp:11-31 with sigma_x = 0 (vectorised); std_phi swaps in a proper CDF."""
    from scipy.special import erf, erfc
    sigma2 = s20
    sigma = np.sqrt(sigma2)
    z = y * f / sigma
    if std_phi:
        ef = 0.5 * erfc(-z / np.sqrt(2.0))
    else:
        ef = erf(z) / float(np.float32(2.0) * np.sqrt(np.float32(2.0)))
    efp = np.exp(-z * z / 2) / np.sqrt(2.0 * np.pi)
    first = efp / ef
    return y / sigma * first, (-z * first - first * first) / sigma2


def laplace_mode_rw(X, y, p0, p1, s20, std_phi=True, f_init=0.0, max_iter=20, tol=1e-9):
    """Algorithm 3.1 of Rasmussen & Williams (2006) as printed -- B = I + W^1/2 K W^1/2, b = W f + grad,
    a = b - W^1/2 B^-1 W^1/2 K b, f = K a -- i.e. a DIFFERENT algebraic route to the Newton iterate than the
    (K + W^-1)^-1 form of the oracle and the GPU kernel.  Same start (f = y f_init) and stopping rule.
    Returns fhat, a, iters."""
    from scipy.linalg import cho_factor, cho_solve
    n = X.shape[0]
    K = rbf(p0, p1, X, X)
    f = y * f_init
    a = np.zeros(n)
    it = 0
    for it in range(1, max_iter + 1):
        g, r = probit_functor(y, f, s20, std_phi)
        W = -r
        sW = np.sqrt(W)
        B = np.eye(n) + sW[:, None] * K * sW[None, :]
        cf = cho_factor(B, lower=True)
        b = W * f + g
        a = b - sW * cho_solve(cf, sW * (K @ b))
        fn = K @ a
        delta = np.max(np.abs(fn - f))
        f = fn
        if delta <= tol:
            break
    return f, a, it
