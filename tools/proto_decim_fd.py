"""developer tool: float64 check of the identity behind csrc/fir_decim_fd.hip (decimate-by-8 FIR as one 4096-point complex transform, one table product and a
1024-point inverse transform per 8192-sample overlap-save block)"""
import numpy as np
rng = np.random.default_rng(1)
N, D, K = 8192, 8, 1024
V = 1024                      # overlap (multiple of 8), >= K - 1 rounded up to 8
Hop = N - V
b = (rng.standard_normal(K) * np.hamming(K)).astype(np.float64)
x = rng.standard_normal(5 * Hop + N)
# reference: y[m] = sum_k b[k] x[8m - k]
full = np.convolve(x, b)[: len(x)]
yref = full[::D]
# block j: xb = x[j*Hop - V + (0..N-1)] (zeros before the start)
xe = np.concatenate([np.zeros(V), x])
NC = N // 2                   # 4096 complex points
Hs = np.fft.fft(np.concatenate([b, np.zeros(N - K)]))          # H[k], k = 0..8191
k = np.arange(NC + 1)
W = np.exp(-2j * np.pi * k / N)
# X[k] = alpha[k] Z[k] + beta[k] conj(Z[NC-k]),  alpha = (1 - i W^k)/2, beta = (1 + i W^k)/2
alpha, beta = (1 - 1j * W) / 2, (1 + 1j * W) / 2
P = Hs[: NC + 1] * alpha      # C[k] = P[k] Z[k] + Q[k] conj Z[NC - k], k = 0..NC
Q = Hs[: NC + 1] * beta
# R[k] = P[k] + conj(Q[NC - k]) for k = 0..NC-1 (Z periodic: Z[NC] = Z[0])
R = P[:NC] + np.conj(Q[NC - np.arange(NC)])
R4096 = P[NC] + np.conj(Q[0])
Rm = R.copy()
Rm[0] = (R[0] + R4096) / 2
out = []
for j in range(5):
    xb = xe[j * Hop: j * Hop + N]
    z = xb[0::2] + 1j * xb[1::2]
    Z = np.fft.fft(z)
    # G[r] = sum_{jj<4} Rm[r + 1024 jj] Z[r + 1024 jj], r = 0..1023
    G = (Rm * Z).reshape(4, 1024).sum(axis=0)
    g = np.fft.ifft(G) * 1024 / 1024          # ifft includes 1/1024
    d = np.real(g) / 4 * 1.0                   # d[i'] = c[8 i']
    out.append(d[V // D:])                     # valid outputs
got = np.concatenate(out)
print("max err", np.max(np.abs(got - yref[: len(got)])), "scale", np.max(np.abs(yref)))
