"""Timing of the f64 whitening-learning kernels (mdx_gram_f64, mdx_project_f64) at D = 2048, n = 20 000 next to
torch.matmul (rocBLAS dgemm); roofline against the f64 matrix peak (78.6 TFLOP/s: AMD's MI355X figure, matrix = vector)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mdir_amd import ops

PEAK_F64_TFLOPS = 78.6
dev = "cuda:0"
D, n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
g = torch.Generator(device=dev); g.manual_seed(0)
A = torch.randn((D, n), generator=g, device=dev, dtype=torch.float64)
P = torch.randn((D, D), generator=g, device=dev, dtype=torch.float64)
m = torch.randn(D, generator=g, device=dev, dtype=torch.float64)


def timed(fn, reps=5):
    fn(); fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


t_gram = timed(lambda: ops.gram_f64(A))
t_gram_blas = timed(lambda: A @ A.t())
t_proj = timed(lambda: ops.project_f64(P, A, m))
t_proj_blas = timed(lambda: P @ (A - m[:, None]))
tile = 128 if D >= 1024 else 64                                            # gram_tile() of csrc/mdx_gram.hip
tiles = -(-D // tile)
flops_gram_done = 2.0 * n * tile * tile * (tiles * (tiles + 1) // 2)      # tiles on or above the diagonal
flops_proj = 2.0 * D * D * n
err = float((ops.gram_f64(A) - A @ A.t()).abs().max() / (A @ A.t()).abs().max())
out = {"D": D, "n": n,
       "gram_ms": round(t_gram, 3), "gram_tflops_executed": round(flops_gram_done / t_gram / 1e9, 2),
       "gram_frac_of_f64_peak": round(flops_gram_done / t_gram / 1e9 / PEAK_F64_TFLOPS, 4),
       "gram_equivalent_gemm_tflops": round(2.0 * D * D * n / t_gram / 1e9, 2), "gram_rocblas_ms": round(t_gram_blas, 3),
       "project_ms": round(t_proj, 3), "project_tflops": round(flops_proj / t_proj / 1e9, 2),
       "project_frac_of_f64_peak": round(flops_proj / t_proj / 1e9 / PEAK_F64_TFLOPS, 4), "project_rocblas_ms": round(t_proj_blas, 3),
       "gram_max_rel_diff_vs_rocblas": err}
print(json.dumps(out))
