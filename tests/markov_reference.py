"""Numerically exact acceptance and lag-k autocorrelation of the random-walk Metropolis chain of metropolis.hpp:28-50 /
Metropolis.cuh:40-52 on a 1-D density, from the transition OPERATOR instead of from samples (test infrastructure).

State space [0, L] cut into N cells of width h (midpoints x_i); proposal y = x + s z, z ~ N(0, 1):
    P_ij = phi((x_j - x_i) / s) / s * min(1, p(x_j) / p(x_i)) * h      (j != i)
    P_ii = 1 - sum_{j != i} P_ij                                        (rejections, and proposals onto r < 0 where p = 0)
pi_i ~ p(x_i) satisfies detailed balance with this P exactly (pi_i P_ij = phi h / s * min(p_i, p_j) is symmetric), so the
discrete chain is a reversible chain in its own right and converges to the continuous one as h^2.  Then, with
xt = x - <x>:   rho_k = <xt, P^k xt>_pi / <xt, xt>_pi,   acceptance = sum_i pi_i sum_j phi h / s min(1, p_j / p_i),
tau_int = 1 + 2 sum_{k <= K} rho_k  (the efficiency of StepSizeAnalysis.ipynb cell 1 is 1 / tau_int with K = 49)."""
import numpy as np


def density(kind, p0, p1=0.0):
    """The callables of Distributions.cuh:7-28 on r >= 0 (they are 0 for r < 0)."""
    return {"ws": lambda r: 1.0 / (1.0 + np.exp((r - p0) / p1)),
            "r2ws": lambda r: r * r / (1.0 + np.exp((r - p0) / p1)),
            "gauss": lambda r: np.exp(-r * r / (2.0 * p0)),
            "r2gauss": lambda r: r * r * np.exp(-r * r / (2.0 * p0))}[kind]


def chain_statistics(p, step, upper, n_cells=3000, n_lags=49):
    """-> dict(acceptance, mean, variance, rho[n_lags], tau_int) of the chain with proposal width `step` on density p."""
    h = upper / n_cells
    x = (np.arange(n_cells) + 0.5) * h
    px = p(x)
    pi = px / px.sum()
    d = (x[None, :] - x[:, None]) / step
    P = np.exp(-0.5 * d * d) * (h / (step * np.sqrt(2.0 * np.pi)))
    with np.errstate(divide="ignore", invalid="ignore"):
        P *= np.minimum(1.0, px[None, :] / px[:, None])
    np.fill_diagonal(P, 0.0)
    moved = P.sum(axis=1)
    # a proposal that lands in its own cell is an accepted one (alpha = 1 there): its mass phi(0) h / s belongs to the
    # acceptance although it does not move the discrete chain
    acceptance = float(pi @ moved) + h / (step * np.sqrt(2.0 * np.pi))
    P[np.diag_indices(n_cells)] = 1.0 - moved
    mean = float(pi @ x)
    xt = x - mean
    var = float(pi @ (xt * xt))
    rho = np.empty(n_lags)
    v = xt.copy()
    for k in range(n_lags):
        v = P @ v
        rho[k] = float(pi @ (xt * v)) / var
    return {"acceptance": acceptance, "mean": mean, "variance": var, "rho": rho, "tau_int": 1.0 + 2.0 * float(rho.sum())}


def richardson(p, step, upper, n_cells=2000, n_lags=49):
    """chain_statistics at n_cells and 2 n_cells, extrapolated in h^2 -> (extrapolated dict, largest |change| of a rho_k
    between the fine grid and the extrapolation: the size of the discretisation error that is left)."""
    a = chain_statistics(p, step, upper, n_cells, n_lags)
    b = chain_statistics(p, step, upper, 2 * n_cells, n_lags)
    out = {}
    for k in a:
        out[k] = b[k] + (b[k] - a[k]) / 3.0
    return out, float(np.max(np.abs(out["rho"] - b["rho"])))
