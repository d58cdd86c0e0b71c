# Fit the odd polynomial sin r = r + r z P(z), z = r^2, on |r| <= pi/2 for f32 evaluation; weighted least squares at
# Chebyshev nodes in high precision (relative error), then rounding to f32 and a float32-arithmetic check.
import mpmath as mp, numpy as np
mp.mp.dps = 50
def fit(ncoef, half=mp.pi/2):
    N = 400
    xs = [half * mp.cos(mp.pi * (2*i+1) / (2*N)) for i in range(N)]
    xs = [x for x in xs if x > 0]
    A = mp.matrix(len(xs), ncoef); b = mp.matrix(len(xs), 1)
    for i, r in enumerate(xs):
        z = r*r
        # (sin r - r) / (r z) = P(z); weight so that relative error of sin is minimised: multiply by r z / sin r
        w = r*z / mp.sin(r)
        for j in range(ncoef): A[i, j] = w * z**j
        b[i] = w * (mp.sin(r) - r) / (r*z)
    c = mp.lu_solve(A, b)
    return [c[j] for j in range(ncoef)]
for nc in (4, 5):
    c = fit(nc)
    cf = [np.float32(float(x)) for x in c]
    print(nc, [float(x) for x in c]); print('   f32:', [repr(float(x)) for x in cf])
    # check in float32 arithmetic (fma emulated through float64)
    rs = np.linspace(-np.pi/2, np.pi/2, 400001).astype(np.float32)
    def fma(a, b, c_): return (a.astype(np.float64) * b.astype(np.float64) + c_.astype(np.float64)).astype(np.float32)
    z = (rs.astype(np.float64) * rs.astype(np.float64)).astype(np.float32)
    p = np.full_like(rs, cf[-1])
    for k in range(nc - 2, -1, -1): p = fma(p, z, np.full_like(rs, cf[k]))
    rz = (rs.astype(np.float64) * z.astype(np.float64)).astype(np.float32)
    v = fma(rz, p, rs)
    ref = np.sin(rs.astype(np.float64))
    ulp = np.abs(np.spacing(ref.astype(np.float32)))
    err = np.abs(v.astype(np.float64) - ref) / np.maximum(ulp, 1e-45)
    print('   max ulp err', err.max(), 'at r=', rs[err.argmax()], 'rel', np.max(np.abs(v - ref) / np.maximum(np.abs(ref), 1e-30)))
