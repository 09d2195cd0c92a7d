import torch, time
dev = torch.device("cuda:0")
V = 16667
A = torch.zeros(V, V, dtype=torch.float64, device=dev)
idx = torch.arange(V, device=dev)
A[idx, idx] = 8.0
for sft in (1, 2, 129):
    A[idx[:-sft], idx[sft:]] = -1.0; A[idx[sft:], idx[:-sft]] = -1.0
def t(f, n=2):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, r
dt, Lc = t(lambda: torch.linalg.cholesky(A)); print(f"cholesky {V}: {dt:.3f} s", flush=True)
dt, _ = t(lambda: A @ A); print(f"fp64 gemm {V}^3: {dt:.3f} s = {2*V**3/dt/1e12:.1f} TFLOP/s", flush=True)
def tri_inv(L, nb=1024):
    n = L.shape[0]
    if n <= nb:
        return torch.linalg.solve_triangular(L, torch.eye(n, dtype=L.dtype, device=L.device), upper=False)
    h = (n // 2 + 255) // 256 * 256
    A11 = tri_inv(L[:h, :h], nb); A22 = tri_inv(L[h:, h:], nb)
    out = torch.zeros_like(L)
    out[:h, :h] = A11; out[h:, h:] = A22
    out[h:, :h] = -(A22 @ (L[h:, :h] @ A11))
    return out
dt, Li = t(lambda: tri_inv(Lc), 1); print(f"blocked triangular inverse: {dt:.3f} s", flush=True)
dt, Ai = t(lambda: Li.mT @ Li, 1); print(f"Li^T Li: {dt:.3f} s", flush=True)
print("resid", float((A @ Ai[:, :512] - torch.eye(V, dtype=torch.float64, device=dev)[:, :512]).abs().max()))
try:
    dt, X = t(lambda: torch.linalg.inv(A), 1); print(f"linalg.inv: {dt:.3f} s")
except Exception as e:
    print("inv failed", str(e)[:100])
