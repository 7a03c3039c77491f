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


@pytest.mark.parametrize("n_images", [90, 1125])
def test_configs4_end_to_end_vgg16_multiscale_whitening_fp16(tmp_path, n_images):
    """BASELINE.json configs[4] through the product's scenario surface: VGG16-GeM (random init), 3 scales + learned
    whitening (learned here from the set's descriptors; the cirwhiten / cirmultiscale wrapper chain), a 247tokyo1k-shaped set (query == database, the image
    itself in `junk`: cirscore.py:56-57), descriptors kept as fp16 (`criterion: {storage: f16}`, scenarios/eval_fp16.yml)
    against the fp32 shard.  Stated bounds: |mAP(fp16) - mAP(fp32)| <= 0.005; the two top-10 lists name the same ids in
    >= 97 % of the slots; every fp16 score within 2e-3 of the fp32 one.  n_images = 1125: the real set's size (N = Q = 1125,
    D = 512: the shape of the bench line's `configs4_247tokyo1k_shape` leg, here with extracted descriptors)."""
    import json
    import os
    import subprocess
    import sys
    import yaml
    from conftest import ROOT
    from mdir_amd import ops, stages
    from mdir_amd.datasets import configdataset, initialize_transforms
    from mdir_amd.network import load_network
    from mdir_amd.networks import extract_vectors_device
    from mdir_amd.scenario import dict_deep_overlay
    root = str(tmp_path / "synth")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_synthetic_eval.py"), root, "vgg16", str(n_images)])
    os.environ["CIRTORCH_ROOT"] = root
    os.environ["MDIR_AMD_WORKERS"] = "2"

    def scenario(*overlays):
        sc = {}
        for name in ("eval.yml",) + overlays:
            path = name if os.path.isabs(name) else os.path.join(ROOT, "scenarios", name)
            sc = dict_deep_overlay(sc, yaml.safe_load(open(path)))
        sc["validation"].pop("roxford5k")               # configs[4] is the Tokyo 24/7 set
        return sc

    # "learned whitening": PCA whitening (Arun's shrinkage) learned from the set's own multi-scale descriptors with the
    # product's learner (whitenlearn.py:14-35 on mdx_gram_f64) -- a random-init trunk puts every image within 1e-3 of
    # every other one, and only a whitening that removes the common mean makes this a retrieval problem
    import pickle
    from mdir_amd.whiten import pcawhitenlearn
    raw = scenario(os.path.join(root, "eval_synth.yml"))
    raw["network"]["runtime"]["wrappers"]["eval"].pop("0_cirwhiten")
    cfg = configdataset("247tokyo1k", os.path.join(root, "data", "test"))
    images = [cfg["im_fname"](cfg, i) for i in range(cfg["n"])]
    net_raw = load_network(raw["network"], torch.device(DEV)).eval()
    tr = initialize_transforms("pil2np | totensor | normalize", net_raw.network_params.runtime["data"]["mean_std"])
    with torch.no_grad():
        X = extract_vectors_device(net_raw, images, 320, tr, device=torch.device(DEV)).cpu().numpy().astype(np.float64).T
    m, P = pcawhitenlearn(X, shrink=32, device=DEV)
    with open(os.path.join(root, "whiten.pkl"), "wb") as f:
        pickle.dump({"m": m, "P": np.real(P)}, f)

    key = "247tokyo1k/validation/score:ap_avg.4"
    map32 = stages.validate(scenario(os.path.join(root, "eval_synth.yml")), ())[0]["eval"][key]
    map16 = stages.validate(scenario(os.path.join(root, "eval_synth.yml"), "eval_fp16.yml"), ())[0]["eval"][key]
    # the same descriptors, both shards: score and top-10 agreement
    sc = scenario(os.path.join(root, "eval_synth.yml"))
    net = load_network(sc["network"], torch.device(DEV)).eval()
    with torch.no_grad():
        vecs = extract_vectors_device(net, images, 320, tr, device=torch.device(DEV))       # [N, 512]
    assert vecs.shape == (n_images, 512)
    s32 = ops.DescriptorIndex(vecs, "ND").scores(vecs, "ND")
    s16 = ops.DescriptorIndex(vecs, "ND", storage="f16").scores(vecs, "ND")
    worst = float((s32 - s16).abs().max())
    t32, _ = ops.topk(s32, 10)
    t16, _ = ops.topk(s16, 10)
    agree = float((t32 == t16).float().mean())
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        json.dump({"map_fp32": map32, "map_fp16": map16, "max_abs_score_diff": worst, "top10_slot_agreement": agree},
                  open(os.path.join(out, "configs4_measured_%d.json" % n_images), "w"))
    assert 0.05 < map32 < 0.999, map32                                   # a non-trivial retrieval problem
    assert abs(map16 - map32) <= 0.005, (map16, map32)
    assert worst <= 2e-3, worst
    assert agree >= 0.97, agree
