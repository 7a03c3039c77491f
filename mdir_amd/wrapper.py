"""Inference wrappers: multi-scale aggregation and learned whitening.

Drop-in for ``mdir/components/data/wrapper.py``: ``Compose`` (:8-41), ``Wrapper``
(:44-57), ``CirMultiscaleAggregation`` (:84-136), ``CirtorchWhiten`` (:181-195),
``WRAPPERS_LABELS`` / ``initialize_wrappers`` (:198-220).  Same registry keys, same
constructor arguments, same preprocess/postprocess protocol and ordering
(pre-processing in list order, post-processing reversed).  The arithmetic is
``mdx_ms_aggregate`` and ``mdx_scores`` + ``mdx_l2n_rows``; the whitening matrix is
re-tiled ONCE into a resident shard instead of being re-read as a torch mat-vec for
every image.  FakeBatch / CirFakeTupleBatch / ReflectPadMakeDivisible belong to
training and the U-Net pre-networks and are out of scope.
"""
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .graphs import parallel_map


def load_path(path):
    """Pickle from a local path or a (mirrored, hash-checked) URL: mdir/tools/utils.py:44-51."""
    assert path.endswith(".pkl"), "Cannot load anything else than pickle at the moment"
    from .scenario import open_resource
    return pickle.load(open_resource(path))


class Compose(object):
    def __init__(self, wrappers, device):
        self.wrappers = wrappers
        self.device = device

    def __call__(self, tensor, inference, model=None):
        """Pre-process in list order, run ``inference`` on the tensor (or on every tensor of a list),
        post-process in REVERSE order (wrapper.py:17-37)."""
        model = inference if model is None else model

        def run(x):
            out = inference(x.to(self.device))
            # a batch of equal-sized images (the reference is batch-1 only): [D,B] -> one row per image
            if x.dim() == 4 and x.shape[0] > 1 and isinstance(out, torch.Tensor) and out.dim() == 2:
                out = out.t()
            return out

        pending = []                                   # (wrapper, its metadata) in application order
        for step in self.wrappers:
            tensor, meta = step.preprocess(tensor, model)
            pending.append((step, meta))
        out = None
        if isinstance(tensor, list) and pending and hasattr(pending[-1][0], "fused_tail"):
            # pyramid -> descriptor with the whole tail in two launches (None: not applicable, take the general route)
            out = pending[-1][0].fused_tail(tensor, inference, model, pending[-1][1], self.device)
            if out is not None:
                pending.pop()
        if out is None:
            out = parallel_map(run, tensor) if isinstance(tensor, list) else run(tensor)
        while pending:
            step, meta = pending.pop()
            out = step.postprocess(out, model, meta)
        return out

    def defer_final_whitening(self):
        """If the step applied LAST to a descriptor is a whitening (``0_cirwhiten`` of eval.yml: post-processing runs in
        reverse order), switch it off in this chain and return it: a caller that collects many descriptors
        (``extract_vectors_device``) then whitens the finished ``[N,D]`` matrix ONCE -- one pass over ``P`` instead of
        one per batch of 4 images.  Row for row the same arithmetic (every row is its own k-ordered chain).  Returns
        ``None`` when there is nothing to defer; ``restore_whitening`` puts the step back."""
        if self.wrappers and isinstance(self.wrappers[0], CirtorchWhiten) and not self.wrappers[0].skip:
            self.wrappers[0].skip = True
            return self.wrappers[0]
        return None

    @staticmethod
    def restore_whitening(step):
        if step is not None:
            step.skip = False

    def __repr__(self):
        body = "".join("\n    %s" % w for w in self.wrappers)
        return "%s([%s])" % (type(self).__name__, body + "\n" if body else "")


class Wrapper(object):
    def __init__(self, device):
        pass

    def preprocess(self, tensor, _model):
        return tensor, None

    def postprocess(self, tensor, _model, _metadata):
        return tensor


class CirMultiscaleAggregation(Wrapper):
    """Evaluate the network on an image pyramid and aggregate the descriptors."""

    def __init__(self, scales, device):
        super().__init__(device)
        if isinstance(scales, str):
            scales = {"True": True, "False": False}[scales]
        if isinstance(scales, bool):
            scales = [1, 1. / np.sqrt(2), 1. / 2] if scales else [1]
        self.scales = scales

    def _pyramid(self, tensor):
        # torch >= 1.6 semantics: output size floor(in * s), coordinates scaled by s; on the device all
        # levels come from one launch of the library (mdx_bilinear_pyramid), elsewhere from F.interpolate itself
        if tensor.is_cuda and tensor.dtype == torch.float32 and len(self.scales) <= 8:
            return ops.bilinear_pyramid(tensor.contiguous(), [float(s) for s in self.scales])
        return [F.interpolate(tensor, scale_factor=scale, mode="bilinear", align_corners=False)
                for scale in self.scales]

    def preprocess(self, tensor, _model):
        if len(self.scales) == 1:
            return tensor if isinstance(tensor, list) else [tensor], isinstance(tensor, list)
        if isinstance(tensor, list):
            acc = []
            for single in tensor:
                acc += self._pyramid(single)
            return acc, True
        return self._pyramid(tensor), False

    @staticmethod
    def aggregate_tensor(tensor, nscales, outputdim, msp):
        assert len(tensor) == nscales, "%s != %s" % (len(tensor), nscales)
        if tensor[0].dim() == 2 and tensor[0].shape[0] > 1 and tensor[0].shape[1] == outputdim:
            return ops.ms_aggregate_batch([t.contiguous() for t in tensor], msp)       # batch: S x [B,D] -> [B,D], one launch
        flat = [t.reshape(-1).contiguous() for t in tensor]
        assert all(t.numel() == outputdim for t in flat)
        return ops.ms_aggregate(flat, msp)

    def _msp(self, model):
        """Exponent of the power mean: the network's GeM ``p`` iff several scales, GeM pooling, no regional pooling and
        no in-network whitening (wrapper.py:122-124), else 1."""
        if len(self.scales) > 1 and model.meta["pooling"] == "gem" and not model.meta["regional"] \
                and not model.meta["whitening"]:
            return model.pool.p_value() if hasattr(model.pool, "p_value") else model.pool.p.item()
        return 1

    def fused_tail(self, pyramid, inference, model, waslist, device):
        """``features`` of every scale, then the whole descriptor tail in TWO launches: ``mdx_pool_multi`` (all scales'
        maps) and ``mdx_l2n_aggregate`` (per-scale L2N + power mean + renormalisation) instead of 2 launches per scale
        + 1.  Only when the network is called directly and everything after its ``features`` is pooling + L2N
        (``ImageRetrievalNet.fusable_tail``); bit-identical to the general route.  ``None`` otherwise."""
        spec = model.fusable_tail() if inference is model and hasattr(model, "fusable_tail") else None
        if spec is None or waslist or len(self.scales) < 2 or len(pyramid) != len(self.scales) \
                or os.environ.get("MDIR_AMD_FUSED_TAIL", "1") == "0":
            return None
        feats = parallel_map(lambda x: model.features(x.to(device)).contiguous(), pyramid)
        out = ops.l2n_aggregate(ops.pool_multi(feats, *spec), model.norm.eps, self._msp(model))
        return out if out.shape[0] > 1 else out.reshape(-1)

    def postprocess(self, tensor, model, waslist):
        msp = self._msp(model)
        n = len(self.scales)
        if not waslist:
            return self.aggregate_tensor(tensor, n, model.meta["out_channels"], msp)
        assert len(tensor) % n == 0, "%s %% %s != 0" % (len(tensor), n)
        return [self.aggregate_tensor(tensor[i:i + n], n, model.meta["out_channels"], msp)
                for i in range(0, len(tensor), n)]

    def __repr__(self):
        return "%s(scales=%s)" % (self.__class__.__name__, self.scales)


class CirtorchWhiten(Wrapper):
    """Whiten descriptors with optional dimensionality reduction: ``P[:d] (v - m)``,
    then ``/(||.|| + 1e-6)``."""

    def __init__(self, whitening, dimensions, device):
        super().__init__(device)
        whitening = load_path(whitening) if isinstance(whitening, str) else whitening
        P = torch.tensor(np.asarray(whitening["P"]), dtype=torch.float32, device=device)
        self.m = torch.tensor(np.asarray(whitening["m"]), dtype=torch.float32, device=device).reshape(-1).contiguous()
        self.dimensions = dimensions or P.shape[0]
        self.shard = ops.DescriptorIndex(P[:self.dimensions].contiguous(), "ND")   # resident, re-tiled once
        self.P = P
        self.skip = False                  # Compose.defer_final_whitening: the caller whitens all rows at the end

    def whiten_rows(self, vecs_nd):
        """``[n, D]`` device descriptors -> ``[n, d]`` whitened rows (batched form)."""
        y = self.shard.scores(vecs_nd.contiguous(), "ND", center=self.m)
        return ops.l2n_rows_(y, eps=1e-6)

    def postprocess(self, tensor, model, _meta):
        if self.skip:
            return tensor
        if isinstance(tensor, list):
            return [self.postprocess(t, model, _meta) for t in tensor]
        if tensor.dim() == 2 and tensor.shape[0] > 1 and tensor.shape[1] == self.P.shape[1]:
            return self.whiten_rows(tensor)                     # batch: one row per image
        return self.whiten_rows(tensor.reshape(1, -1)).reshape(-1)

    def __repr__(self):
        return "%s(dimensions=%s)" % (self.__class__.__name__, self.dimensions)


WRAPPERS_LABELS = {
    "cirmultiscale": CirMultiscaleAggregation,
    "cirwhiten": CirtorchWhiten,
}


def _from_string(spec, device):
    """``"name[:arg],name2[:arg]"`` -> wrapper objects; the argument arrives as a string."""
    out = []
    for piece in filter(None, spec.split(",")):
        name, colon, arg = piece.partition(":")
        out.append(WRAPPERS_LABELS[name](*([arg] if colon else []), device=device))
    return out


def initialize_wrappers(net_wrappers, device):
    """``None`` | ``"name:arg,name2"`` | ``{"<order>_<name>": kwargs}`` (applied in sorted key order)."""
    if not net_wrappers:
        return Compose([], device)
    if isinstance(net_wrappers, str):
        return Compose(_from_string(net_wrappers, device), device)
    ordered = sorted(net_wrappers)
    return Compose([WRAPPERS_LABELS[key.split("_", 1)[1]](device=device, **net_wrappers[key]) for key in ordered], device)
