"""fp16 shards (BASELINE.json configs[4]: VGG16-GeM 512-d descriptors, fp16 on the fp16 MFMA).

Looser contract than the fp32 path, stated here:
  * scores = sum_k fp16(q_k) * fp16(v_k) with fp32 accumulation: every product is exact in fp32,
    so against a float64 dot product of the fp16-ROUNDED inputs the error is accumulation rounding
    only -> atol 2e-6;
  * against the fp32 reference scores the difference is the input rounding: <= 2e-3 for unit vectors;
  * the ranking is exactly the stable descending order of the GPU's own scores (integer work),
    and agrees with the fp32 ranking wherever fp32 scores differ by more than the rounding bound."""
import numpy as np
import pytest
import torch

from oracle import chain as OC
from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("n,d,nq", [(75984, 512, 315), (1125, 512, 1125), (70000, 2048, 70), (333, 100, 17)])
def test_f16_scores_and_ranking(n, d, nq):
    from mdir_amd import ops
    vecs, qvecs, qid = O.synth_ranking_problem(n, min(nq, n), d, seed=11)
    nq = qvecs.shape[1]
    ix = ops.DescriptorIndex(dev(vecs), "DN", storage="f16")
    assert ix.device_bytes <= (n + 128) * (-(-d // 64) * 64) * 2         # half the bytes of an fp32 shard
    sc = ix.scores(dev(qvecs), "DN")
    got = sc.cpu().numpy()
    v16 = vecs.astype(np.float16).astype(np.float64)
    q16 = qvecs.astype(np.float16).astype(np.float64)
    np.testing.assert_allclose(got, (q16.T @ v16), rtol=0, atol=2e-6)
    np.testing.assert_allclose(got.T, O.scores(vecs, qvecs), rtol=0, atol=2e-3)
    rk = ops.rank_full(sc).cpu().numpy()
    np.testing.assert_array_equal(rk, OC.rank_full(got))                 # exact order of its own scores
    assert (rk[:, 0] == qid).mean() > 0.99                               # queries still find their source row
    # row-major input gives the same bits
    ix2 = ops.DescriptorIndex(dev(np.ascontiguousarray(vecs.T)), "ND", storage="f16")
    np.testing.assert_array_equal(ix2.scores(dev(np.ascontiguousarray(qvecs.T)), "ND").cpu().numpy(), got)


def test_f16_whitening_projection():
    """P[:d] (v - m) through an fp16 shard of P: centred in fp32, then rounded."""
    from mdir_amd import ops
    rng = np.random.default_rng(1)
    P = (rng.standard_normal((128, 512)) / 16).astype(np.float32)
    X = rng.standard_normal((40, 512)).astype(np.float32)
    m = rng.normal(0, 0.01, 512).astype(np.float32)
    ix = ops.DescriptorIndex(dev(P), "ND", storage="f16")
    got = ix.scores(dev(X), "ND", center=dev(m)).cpu().numpy()
    want = (X - m).astype(np.float16).astype(np.float64) @ P.astype(np.float16).astype(np.float64).T
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
