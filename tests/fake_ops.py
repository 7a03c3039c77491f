"""Oracle-backed stand-ins for mdir_amd.ops, for the CPU-only host-logic tests.

TEST-ONLY: the product never imports this.  `install(monkeypatch)` swaps the functions
of `mdir_amd.ops` that launch HIP kernels for CPU equivalents computed by the oracle, so
that wrappers / networks / scores / scenarios can be exercised without a GPU.  The same
host code is run against the real library in the -m gpu tests.
"""
import numpy as np
import torch

from oracle import chain as OC
from oracle import oracle as O


def pool_l2n(feat, kind="gem", p=3.0, pool_eps=1e-6, l2n_eps=1e-6):
    x = feat.detach().numpy()
    pooled = {"gem": lambda: O.gem(x, p, pool_eps), "mac": lambda: O.mac(x), "spoc": lambda: O.spoc(x)}[kind]()
    if l2n_eps is not None:
        pooled = O.l2n(pooled, l2n_eps)
    return torch.from_numpy(np.ascontiguousarray(pooled))


def rmac(feat, regions, eps=1e-6):
    """The product's region list (whole map first) through the oracle's arithmetic: sum of the L2-normalised region maxima."""
    x = feat.detach().numpy()
    v = None
    for i0, j0, h, w in regions:
        t = O.l2n(x[:, :, i0:i0 + h, j0:j0 + w].reshape(x.shape[0], x.shape[1], -1).max(axis=2), eps)
        v = t if v is None else (v + t).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(v))


def roipool(feat, regions, kind="gem", p=3.0, pool_eps=1e-6):
    x = feat.detach().numpy()
    fn = {"gem": lambda a: O.gem(a, p, pool_eps), "mac": O.mac, "spoc": O.spoc}[kind]
    out = np.stack([fn(np.ascontiguousarray(x[:, :, i0:i0 + h, j0:j0 + w])) for i0, j0, h, w in regions], axis=1)
    return torch.from_numpy(np.ascontiguousarray(out.astype(np.float32)))


def region_sum(vecs, l2n_eps=None):
    v = vecs.detach().numpy()
    if l2n_eps is not None:
        v = np.stack([O.l2n(v[:, r], l2n_eps) for r in range(v.shape[1])], axis=1)
    return torch.from_numpy(np.ascontiguousarray(v.sum(axis=1, dtype=np.float32)))


def l2n_cols_f64_(x, eps=1e-6):
    v = x.numpy()
    v /= (np.linalg.norm(v, ord=2, axis=0, keepdims=True) + eps)
    return x


def l2n_rows_(x, bias=None, eps=1e-6):
    v = x.detach().numpy()
    if bias is not None:
        v = v + bias.detach().numpy()[None, :]
    nrm = np.sqrt(np.sum(v * v, axis=1, keepdims=True, dtype=np.float32), dtype=np.float32)
    x.copy_(torch.from_numpy((v / (nrm + np.float32(eps))).astype(np.float32)))
    return x


def ms_aggregate(vecs, msp=1.0):
    return torch.from_numpy(O.ms_aggregate(np.stack([v.detach().numpy().reshape(-1) for v in vecs]), msp))


def ms_aggregate_batch(mats, msp=1.0):
    st = np.stack([m.detach().numpy() for m in mats])            # [S,B,D]
    return torch.from_numpy(np.stack([O.ms_aggregate(st[:, b], msp) for b in range(st.shape[1])]))


def pool_multi(feats, kind="gem", p=3.0, pool_eps=1e-6):
    return torch.stack([pool_l2n(f, kind, p, pool_eps, l2n_eps=None) for f in feats])


def l2n_aggregate(pooled, l2n_eps=1e-6, msp=1.0):
    st = np.stack([O.l2n(pooled[s].detach().numpy(), l2n_eps) for s in range(pooled.shape[0])])     # [S,B,D]
    return torch.from_numpy(np.stack([O.ms_aggregate(st[:, b], msp) for b in range(st.shape[1])]))


def resample_u8(images, axis, bounds, taps):
    x, b, k = images.numpy().astype(np.int64), bounds.numpy(), taps.numpy().astype(np.int64)
    x = np.moveaxis(x, 2 if axis == 1 else 1, 0)                 # resampled axis first
    out = np.empty((len(b),) + x.shape[1:], dtype=np.uint8)
    for o, (lo, cnt) in enumerate(b):
        out[o] = np.clip((np.tensordot(k[o, :cnt], x[lo:lo + cnt], axes=(0, 0)) + (1 << 21)) >> 22, 0, 255)
    return torch.from_numpy(np.ascontiguousarray(np.moveaxis(out, 0, 2 if axis == 1 else 1)))


def jpeg_pixels(item, device):
    """Stand-in for mdir_amd.jpeg.pixels: the oracle's restatement of libjpeg on the item's coefficients."""
    info = item.info
    meta = {k: (list(getattr(info, k)) if k.endswith(("samp", "_w", "_h", "offset")) else getattr(info, k))
            for k in ("width", "height", "ncomp", "hsamp", "vsamp", "blocks_w", "blocks_h", "block_offset")}
    rgb = O.jpeg_pixels(item.coef.numpy(), item.quant.numpy().view(np.uint16), meta)[None]
    if item.box:
        x1, y1, x2, y2 = (int(v) for v in item.box)
        rgb = rgb[:, y1:y2, x1:x2]
    return torch.from_numpy(np.ascontiguousarray(rgb))


class DescriptorIndex:
    def __init__(self, vecs, layout="DN", row_offset=0, storage="f32"):
        v = vecs.detach().numpy()
        self.storage = storage
        self.nd = np.ascontiguousarray(v.T if layout in ("DN", "dim_major") else v)
        if storage == "f16":                # shard (and queries) rounded to fp16, fp32 accumulation
            self.nd = self.nd.astype(np.float16).astype(np.float32)
        self.n, self.d = self.nd.shape
        self.row_offset = row_offset
        self.device = vecs.device

    def scores(self, queries, qlayout="DN", center=None, out=None, compute="chain"):
        q = queries.detach().numpy()
        q = np.ascontiguousarray(q.T if qlayout in ("DN", "dim_major") else q)
        if center is not None:
            q = q - center.detach().numpy().reshape(1, -1)
        if self.storage == "f16":
            q = q.astype(np.float16).astype(np.float32)
        res = torch.from_numpy(OC.gemm_nt_chain(q, self.nd))
        if out is not None:
            out.copy_(res)
            return out
        return res

    def close(self):
        pass


def scores_rowmajor(db, queries, qlayout="DN", center=None, out=None):
    return DescriptorIndex(db, "ND").scores(queries, qlayout, center=center, out=out)


def rank_full(scores, id_offset=0, out=None, workspace=None):
    return torch.from_numpy(OC.rank_full(scores.detach().numpy()) + id_offset)


def topk(scores, k, id_offset=0, workspace=None):
    rk = OC.rank_full(scores.detach().numpy())[:, :k]
    return torch.from_numpy(rk + id_offset), torch.from_numpy(np.take_along_axis(scores.numpy(), rk, axis=1))


def rank_of(scores, id_lists):
    s = scores.detach().numpy()
    off, pos, sc = [0], [], []
    for q, ids in enumerate(id_lists):
        ids = np.asarray(ids, dtype=np.int64)
        pos.append(OC.rank_of(s[q], ids))
        sc.append(s[q][ids])
        off.append(off[-1] + len(ids))
    return (torch.from_numpy(np.concatenate(pos) if pos else np.empty(0, np.int64)),
            torch.from_numpy(np.concatenate(sc).astype(np.float32) if sc else np.empty(0, np.float32)), off)


def gather_scores(scores, ids_t, off_t):
    s, off, ids = scores.detach().numpy(), off_t.numpy(), ids_t.numpy()
    out = np.empty(len(ids), dtype=np.float32)
    for q in range(len(off) - 1):
        out[off[q]:off[q + 1]] = s[q][ids[off[q]:off[q + 1]]]
    return torch.from_numpy(out)


def rank_count_(cnt, scores, id_offset, ref_scores, ref_ids, off_t):
    s, off = scores.detach().numpy(), off_t.numpy()
    for q in range(len(off) - 1):
        keys = np.array([OC.desc_key(x) for x in s[q]], dtype=np.uint64)
        gid = np.arange(s.shape[1]) + id_offset
        for t in range(off[q], off[q + 1]):
            rk, ri = OC.desc_key(float(ref_scores[t])), int(ref_ids[t])
            cnt[t] += int(np.count_nonzero((keys < rk) | ((keys == rk) & (gid < ri))))
    return cnt


def clahe_u8_to_chw(images, clip_limit, grid, mean, std, return_intermediates=False):
    outs = []
    for img in images.numpy():
        rgb, _ = O.apply_clahe_rgb(img, clip_limit, tuple(grid) if isinstance(grid, (tuple, list)) else int(grid))
        outs.append(((rgb - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)).astype(np.float32).transpose(2, 0, 1))
    return torch.from_numpy(np.stack(outs))


def gram_f64(a, center=None):
    x = a.detach().numpy()
    if center is not None:
        x = x - center.detach().numpy().reshape(-1, 1)
    return torch.from_numpy(x @ x.T)


def project_f64(p, x, center=None):
    xv = x.detach().numpy()
    if center is not None:
        xv = xv - center.detach().numpy().reshape(-1, 1)
    return torch.from_numpy(p.detach().numpy() @ xv)


NAMES = ("l2n_cols_f64_", "rmac", "roipool", "region_sum", "clahe_u8_to_chw", "gram_f64", "project_f64", "pool_l2n", "l2n_rows_", "ms_aggregate", "ms_aggregate_batch", "pool_multi", "l2n_aggregate", "resample_u8", "DescriptorIndex", "scores_rowmajor", "rank_full", "topk", "rank_of",
         "gather_scores", "rank_count_")


def install_globally():
    """For worker processes of the multi-process tests (no monkeypatch fixture there)."""
    from mdir_amd import ops
    for name in NAMES:
        setattr(ops, name, globals()[name])


def install(monkeypatch):
    from mdir_amd import ops
    for name in NAMES:
        monkeypatch.setattr(ops, name, globals()[name])
