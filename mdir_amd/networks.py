"""Image-retrieval network and descriptor extraction -- the cirtorch operator API.

Drop-in for ``mdir/external/cirtorch/networks/imageretrievalnet.py``:
``ImageRetrievalNet`` (:82-115), ``init_network`` (:138-274), ``extract_vectors`` /
``extract_ss`` / ``extract_ms`` (:277-324).  The convolutional ``features`` run on
PyTorch-ROCm; everything after them (pool -> L2N -> whitening -> L2N, multi-scale
aggregation) runs in the HIP library.  Differences from the reference, all
MI355X-motivated and result-preserving:

* descriptors are written into ONE device-resident ``[N,D]`` buffer and copied to the
  host once, instead of a blocking ``.cpu()`` per image (imageretrievalnet.py:307);
* the in-network whitening ``nn.Linear`` is applied as a resident ``mdx_index`` of its
  weight (re-tiled once), not a per-image mat-vec through torch;
* nothing is ever downloaded: ``pretrained`` only loads files that already exist
  under ``model_dir`` / ``$CIRTORCH_ROOT/data``.
"""
import os
import pickle

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .backbones import OUTPUT_DIM, TrunkSequential, build_features
from .datasets import ImagesFromList, ToUint8HWC, device_convert, get_data_root, make_loader
from .graphs import ShapeGraphs, graphs_enabled, parallel_map
from .layers import POOLING, L2N, Rpool, pool_kind
from .jpeg import pixels as pixels_of
from .resample import DeviceThumbnail


class ImageRetrievalNet(nn.Module):
    supports_batches = True          # forward answers [D,B]; extract_vectors may batch equal-sized images

    def __init__(self, features, lwhiten, pool, whiten, meta):
        super().__init__()
        self.features = TrunkSequential(*features)
        self.lwhiten = lwhiten
        self.pool = pool
        self.whiten = whiten
        self.norm = L2N()
        self.meta = meta
        self._whiten_index = (None, None)

    def _whiten_shard(self):
        """Resident, re-tiled copy of ``whiten.weight`` ([D_out, D_in] row-major = D_out
        "database rows"); rebuilt only when the parameter changes."""
        w = self.whiten.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if self._whiten_index[0] != key:
            self._whiten_index = (key, ops.DescriptorIndex(w.detach().contiguous(), "ND"))
        return self._whiten_index[1]

    def forward(self, x):
        o = self.features(x)
        if self.lwhiten is not None:   # local whitening: plain torch, not on the eval.py path
            s = o.size()
            o = o.permute(0, 2, 3, 1).contiguous().view(-1, s[1])
            o = self.lwhiten(o)
            o = o.view(s[0], s[2], s[3], self.lwhiten.out_features).permute(0, 3, 1, 2)
        kind = pool_kind(self.pool)
        if kind is not None:           # fused pool + L2N: mdx_pool_l2n
            o = ops.pool_l2n(o.contiguous(), kind[0], kind[1], kind[2], l2n_eps=self.norm.eps)
        else:
            o = self.norm(self.pool(o)).squeeze(-1).squeeze(-1)
        if self.whiten is not None:    # W o + b, then L2N: mdx_scores on the weight shard + mdx_l2n_rows
            y = self._whiten_shard().scores(o.contiguous(), "ND")
            bias = self.whiten.bias.detach() if self.whiten.bias is not None else None
            o = ops.l2n_rows_(y, bias=bias, eps=self.norm.eps)
        return o.permute(1, 0)

    def fusable_tail(self):
        """``(kind, p, eps)`` of the pooling when everything after ``features`` is pooling + L2N (no local / in-network
        whitening, a pooling the library knows): the scales of a pyramid can then share one pooling launch
        (``mdx_pool_multi``) and one ``mdx_l2n_aggregate``.  ``None`` otherwise."""
        if self.lwhiten is not None or self.whiten is not None:
            return None
        return pool_kind(self.pool)

    def meta_repr(self):
        lines = ["  (meta): dict( "]
        for key in ("architecture", "local_whitening", "pooling", "regional", "whitening"):
            lines.append("     {}: {}".format(key, self.meta[key]))
        lines.append("     outputdim: {}".format(self.meta.get("out_channels", self.meta.get("outputdim"))))
        if "mean" in self.meta and "std" in self.meta:
            lines += ["     mean: {}".format(self.meta["mean"]), "     std: {}".format(self.meta["std"])]
        return "\n".join(lines) + "\n  )\n"

    def __repr__(self):
        return super().__repr__()[:-1] + self.meta_repr() + ")"


def _local_file(url_or_path, directory):
    """A file that is already on disk for a reference URL / path, else None."""
    if not url_or_path:
        return None
    if os.path.exists(url_or_path):
        return url_or_path
    cand = os.path.join(directory, os.path.basename(url_or_path))
    return cand if os.path.exists(cand) else None


def init_network(params):
    """Build an ``ImageRetrievalNet`` from the reference's parameter dict
    (architecture, local_whitening, pooling, regional, whitening, mean, std,
    pretrained, model_dir).  ``regional: True`` wraps the pooling into ``Rpool`` with a regional whitening
    ``nn.Linear(dim, dim)`` (imageretrievalnet.py:205-222; its pre-computed weights are a download upstream: random here unless
    a checkpoint's state dict fills ``pool.whiten.*``)."""
    architecture = params.get("architecture", "resnet101")
    local_whitening = params.get("local_whitening", False)
    pooling = params.get("pooling", "gem")
    regional = params.get("regional", False)
    whitening = params.get("whitening", False)
    mean = params.get("mean", [0.485, 0.456, 0.406])
    std = params.get("std", [0.229, 0.224, 0.225])
    pretrained = params.get("pretrained", True)
    model_dir = params.get("model_dir", None)

    if architecture not in OUTPUT_DIM:
        raise ValueError("Unsupported or unknown architecture: {}!".format(architecture))
    if pooling not in POOLING:
        raise KeyError("pooling '%s' is not one of %s" % (pooling, sorted(POOLING)))
    dim = OUTPUT_DIM[architecture]
    features = build_features(architecture)
    lwhiten = nn.Linear(dim, dim, bias=True) if local_whitening else None
    pool = POOLING[pooling]()
    if regional:
        pool = Rpool(pool, nn.Linear(dim, dim, bias=True))

    whiten = None
    if whitening:
        whiten = nn.Linear(dim, dim, bias=True)
        if isinstance(whitening, str):   # pickled {'P','m'}: W = P, b = -P m  (imageretrievalnet.py:229-233)
            with open(whitening, "rb") as handle:
                w = pickle.load(handle)
            P = torch.tensor(w["P"], dtype=torch.float32)
            m = torch.tensor(w["m"], dtype=torch.float32)
            whiten.load_state_dict({"weight": P, "bias": -torch.mm(P, m).squeeze()})

    meta = {"architecture": architecture, "local_whitening": local_whitening, "pooling": pooling,
            "regional": regional, "whitening": whitening, "mean": mean, "std": std, "outputdim": dim}
    net = ImageRetrievalNet(features, lwhiten, pool, whiten, meta)

    if pretrained:
        found = _local_file(params.get("features_file"), model_dir or os.path.join(get_data_root(), "networks"))
        if found:
            net.features.load_state_dict(torch.load(found, map_location="cpu"))
            print(">> {}: features loaded from '{}'".format(os.path.basename(__file__), found))
        else:
            print(">> {}: '{}' built with random weights (no downloads on this path; "
                  "load a checkpoint's state_dict to fill them)".format(os.path.basename(__file__), architecture))
    return net


def _out_dim(net):
    return net.meta["out_channels"] if "out_channels" in net.meta else net.meta["outputdim"]


def _rows(net, x):
    """``net(x)`` as one descriptor row per image of the batch: ``[B, D]``.  Networks answer ``[D,B]``
    (ImageRetrievalNet) or, for a wrapped mdir network, ``[D]`` / ``[B,D]``."""
    out = net(x)
    if out.dim() == 1:
        return out.reshape(1, -1)
    if x.shape[0] > 1:
        return out.t() if isinstance(net, nn.Module) else out        # raw model: [D,B]; wrapped network: rows
    return out.reshape(1, -1)


def extract_ss(net, input):
    """One scale -> device vector ``[D]`` (or ``[B,D]`` for a batch of equal-sized images; no host copy)."""
    rows = _rows(net, input)
    return rows if input.shape[0] > 1 else rows.reshape(-1)


def extract_ms(net, input, ms, msp):
    """One image, several scales -> device vector ``[D]`` (imageretrievalnet.py:309-324)."""
    if input.is_cuda and input.dtype == torch.float32 and len(ms) <= 8:
        pyramid = ops.bilinear_pyramid(input.contiguous(), [float(s) for s in ms])     # all levels in one launch
    else:
        pyramid = [input if s == 1 else F.interpolate(input, scale_factor=s, mode="bilinear", align_corners=False)
                   for s in ms]
    spec = net.fusable_tail() if hasattr(net, "fusable_tail") and os.environ.get("MDIR_AMD_FUSED_TAIL", "1") != "0" else None
    if spec is not None and 2 <= len(pyramid) <= 8:    # the whole tail in two launches, bit-identical to the route below
        feats = parallel_map(lambda x: net.features(x).contiguous(), pyramid)
        out = ops.l2n_aggregate(ops.pool_multi(feats, *spec), net.norm.eps, msp)
        return out if input.shape[0] > 1 else out.reshape(-1)
    per_scale = parallel_map(lambda x: _rows(net, x).contiguous(), pyramid)           # one stream per scale
    agg = [ops.ms_aggregate([rows[b] for rows in per_scale], msp) for b in range(input.shape[0])]
    return torch.stack(agg) if input.shape[0] > 1 else agg[0]


def _gpu_preprocess(device):
    return torch.device(device).type == "cuda" and os.environ.get("MDIR_AMD_GPU_PREPROCESS", "1") != "0"


def _shape_key(images, bbxs, i):
    """Raw size of image i from its file header (no decode), or of its crop box."""
    from PIL import Image
    box = bbxs[i] if bbxs else None
    if box:
        return (int(box[2] - box[0]), int(box[3] - box[1]))
    try:
        with Image.open(images[i]) as handle:
            return tuple(handle.size)
    except Exception:
        return (-1, i)                       # unreadable header: a group of its own (the loader reports the error)


def _same_shape_order(images, bbxs, lo=0, hi=None):
    """Indices ``lo..hi`` of ``images`` ordered so that images of equal raw size (hence equal network
    input shape) are consecutive: every hipGraph is then captured once and replayed for its whole
    group instead of being evicted and re-captured."""
    hi = len(images) if hi is None else hi
    keys = {i: _shape_key(images, bbxs, i) for i in range(lo, hi)}
    return sorted(range(lo, hi), key=lambda i: (keys[i], i))


class ShapeOrder:
    """Sampler form of :func:`_same_shape_order` for long image lists: the list is ordered window by
    window (4096 images) WHILE the loader consumes it, so a million-image extraction does not open
    a million files before its first image.  ``emitted[k]`` is the list index of the k-th item the
    loader yields."""

    WINDOW = 4096

    def __init__(self, images, bbxs):
        self.images, self.bbxs, self.emitted, self.upcoming = images, bbxs, [], []

    def __len__(self):
        return len(self.images)

    def __iter__(self):
        self.emitted, self.upcoming = [], []          # upcoming[k]: images of the same size from item k to the end of its run
        for lo in range(0, len(self.images), self.WINDOW):
            hi = min(len(self.images), lo + self.WINDOW)
            keys = {i: _shape_key(self.images, self.bbxs, i) for i in range(lo, hi)}
            order = sorted(range(lo, hi), key=lambda i: (keys[i], i))
            run_end = len(order)
            left = [0] * len(order)
            for k in range(len(order) - 1, -1, -1):
                if k + 1 < len(order) and keys[order[k + 1]] != keys[order[k]]:
                    run_end = k + 1
                left[k] = run_end - k
            for k, i in enumerate(order):
                self.emitted.append(i)
                self.upcoming.append(left[k])
                yield i


class _Sequential:
    """The caller's order, with the same ``emitted`` bookkeeping as :class:`ShapeOrder`."""

    def __init__(self, n):
        self.n, self.emitted = n, []

    def __len__(self):
        return self.n

    def __iter__(self):
        self.emitted = []
        for i in range(self.n):
            self.emitted.append(i)
            yield i


def batched_loop(loader, order, device, describe, store, missing=None, progress=None, batches=True):
    """Drive ``describe`` over a batch-size-1 loader: consecutive equal-sized images go through the
    network as ONE batch of ``MDIR_AMD_BATCH`` (default 8, then 4 for what is left of a size; only under graph replay, where equal sizes
    have been made consecutive) -- larger GEMMs, fewer launches per image; anything else one by one.
    ``store(index, descriptor)`` receives every result, ``missing(index)`` every unreadable image
    (a loader item that is ``{}``).  ``batches=False`` for a network that only takes one image at a
    time (the reference's protocol; this package's networks declare ``supports_batches``)."""
    bmax = max(1, int(os.environ.get("MDIR_AMD_BATCH", "8"))) if batches and graphs_enabled(device) else 1
    buf = []

    def flush():
        if hasattr(describe, "upcoming"):
            # how many images of this size are still to come (the sampler ordered them): a graph is only captured
            # when enough replays will follow (ShapeGraphs.PAYOFF_IMAGES)
            describe.upcoming = buf[0][2] if buf and buf[0][2] is not None else None
        # whole batches of bmax; what is left of a size goes as one batch of bmax / 2 if there is that much, then one by one
        # (at most three launch shapes -- and graphs -- per image size)
        k = 0
        for width in (bmax, bmax // 2):
            while width > 1 and len(buf) - k >= width:
                rows = describe(torch.cat([t for _, t, _ in buf[k:k + width]], dim=0))
                for (i, _, _), row in zip(buf[k:k + width], rows):
                    store(i, row)
                k += width
        for i, t, _ in buf[k:]:
            store(i, describe(t))
        buf.clear()

    for done, item in enumerate(loader):
        i = order.emitted[done] if hasattr(order, "emitted") else order[done]
        if isinstance(item, dict) and item == {}:
            missing(i)
        else:
            item = pixels_of(item, device) if hasattr(item, "coef") else item.to(device, non_blocking=True)
            if buf and buf[0][1].shape != item.shape:
                flush()
            buf.append((i, item, order.upcoming[done] if getattr(order, "upcoming", None) else None))
            if len(buf) == bmax:
                flush()
        if progress:
            progress(done + 1)
    flush()


def extract_vectors_device(net, images, image_size, transform, bbxs=None, ms=[1], msp=1, print_freq=10,
                           device=None, num_workers=None):
    """Like :func:`extract_vectors` but the result stays on the GPU as ``[N,D]``
    (one descriptor per row), ready to become an index shard."""
    if not device:
        net.cuda()
        device = torch.device("cuda")
    net.eval()
    if num_workers is None:
        num_workers = int(os.environ.get("MDIR_AMD_WORKERS", "8"))
    describe = (lambda x: extract_ss(net, x)) if len(ms) == 1 else (lambda x: extract_ms(net, x, ms, msp))
    # a network whose wrapper chain ends in a whitening (mdir's CirNetwork with 0_cirwhiten): whiten all rows at the end
    chain = getattr(net, "wrappers", {}).get(getattr(net, "stage", None)) if isinstance(getattr(net, "wrappers", None), dict) else None
    final_whitening = chain.defer_final_whitening() if hasattr(chain, "defer_final_whitening") else None
    tail = transform.device_tail() if hasattr(transform, "device_tail") and _gpu_preprocess(device) else None
    resize_on_device = False
    if tail is not None:
        # workers ship uint8 pixels (a quarter of the bytes through shared memory, the pinned copy
        # and PCIe, and no float arithmetic on the host); /255, -mean, /std happen on the GPU -- and so does the
        # LANCZOS down-scale to image_size (Pillow's integer arithmetic, mdx_resample_u8): the workers only decode
        transform, from_tensor = ToUint8HWC(), describe
        resize_on_device = image_size is not None and os.environ.get("MDIR_AMD_GPU_RESIZE", "1") != "0"
        shrink = DeviceThumbnail(image_size) if resize_on_device else (lambda u8: u8)
        convert = device_convert(tail)
        describe = lambda u8: from_tensor(convert(shrink(u8)))
    order = _Sequential(len(images))
    if graphs_enabled(device):
        describe = ShapeGraphs(describe)      # per input shape: eager once, then one hipGraph replay per call
        order = ShapeOrder(images, bbxs)
    # JPEG files: entropy decoding in the loader threads, the rest of the decoder on the device (thread loader only: the
    # coefficients are handed over as an object, not collated)
    decode_on_device = tail is not None and (resize_on_device or image_size is None) and num_workers > 0 \
        and os.environ.get("MDIR_AMD_GPU_JPEG", "1") != "0" \
        and os.environ.get("MDIR_AMD_LOADER", "threads") != "processes"
    dataset = ImagesFromList(root="", images=images, imsize=image_size, bbxs=bbxs, transform=transform,
                             resize_on_device=resize_on_device, decode_on_device=decode_on_device)
    loader = make_loader(dataset, order, num_workers, device)
    state = {"vecs": None}

    def store(i, v):
        if state["vecs"] is None:
            # width from the first descriptor: a dimension-reducing whitening wrapper
            # (cirwhiten dimensions=d) yields d < meta['out_channels'], which the
            # reference's fixed-size buffer (imageretrievalnet.py:291) cannot hold
            state["vecs"] = torch.empty(len(images), v.numel(), dtype=torch.float32, device=device)
        state["vecs"][i].copy_(v.reshape(-1), non_blocking=True)      # row = position in the caller's list

    def progress(done):
        if done % print_freq == 0 or done == len(images):
            print("\r>>>> {}/{} done...".format(done, len(images)), end="")

    try:
        with torch.no_grad():
            batched_loop(loader, order, device, describe, store, progress=progress,
                         batches=getattr(net, "supports_batches", False))
            print("")
    finally:
        if final_whitening is not None:
            chain.restore_whitening(final_whitening)
    vecs = state["vecs"]
    if final_whitening is not None and vecs is not None:
        with torch.no_grad():                  # ONE pass over P for the whole list (blocks bound the fp32 [n,d] temporaries)
            vecs = torch.cat([final_whitening.whiten_rows(vecs[i:i + 65536]) for i in range(0, vecs.shape[0], 65536)], dim=0)
    return vecs


def extract_vectors(net, images, image_size, transform, bbxs=None, ms=[1], msp=1, print_freq=10, device=None):
    """Descriptors of a list of images as a CPU ``torch.float32 [D,N]`` tensor -- the
    reference's signature and return layout (imageretrievalnet.py:277-304)."""
    vecs = extract_vectors_device(net, images, image_size, transform, bbxs, ms, msp, print_freq, device)
    return vecs.t().contiguous().cpu()


# ---------------------------------------------------------------- regional / local descriptors (imageretrievalnet.py:325-384)

def extract_ssr(net, input):
    """The regional vectors of one image, ``[D, R]`` on the host (imageretrievalnet.py:354-355): ``Rpool`` without aggregation."""
    return net.pool(net.features(input), aggregate=False).squeeze(0).squeeze(-1).squeeze(-1).permute(1, 0).cpu().data


def extract_ssl(net, input):
    """The local descriptors of one image, ``[D, H*W]`` on the host (imageretrievalnet.py:383-384): every location of the feature
    map L2-normalised over the channels (``net.norm`` on the map) -- one row per location through ``mdx_l2n_rows``."""
    feat = net.features(input)
    c = feat.shape[1]
    rows = feat.squeeze(0).reshape(c, -1).t().contiguous()              # [H*W, C]
    return ops.l2n_rows_(rows, eps=net.norm.eps).t().contiguous().cpu().data


def _extract_per_image(net, images, image_size, transform, bbxs, ms, print_freq, device, one):
    from .datasets import ImagesFromList, make_loader
    if not device:
        net.cuda()
        device = torch.device("cuda")
    net.eval()
    loader = make_loader(ImagesFromList(root="", images=images, imsize=image_size, bbxs=bbxs, transform=transform), range(len(images)),
                         int(os.environ.get("MDIR_AMD_WORKERS", "8")), device)
    vecs = []
    with torch.no_grad():
        for i, item in enumerate(loader):
            if len(ms) != 1:
                raise NotImplementedError           # as upstream ("TODO: not implemented yet", :344-347, :373-376)
            vecs.append(one(net, item.to(device)))
            if (i + 1) % print_freq == 0 or (i + 1) == len(images):
                print("\r>>>> {}/{} done...".format(i + 1, len(images)), end="")
        print("")
    return vecs


def extract_regional_vectors(net, images, image_size, transform, bbxs=None, ms=[1], msp=1, print_freq=10, device=None):
    """List of ``[D, R_i]`` host tensors, one per image (imageretrievalnet.py:325-352; ``regional: True`` networks)."""
    return _extract_per_image(net, images, image_size, transform, bbxs, ms, print_freq, device, extract_ssr)


def extract_local_vectors(net, images, image_size, transform, bbxs=None, ms=[1], msp=1, print_freq=10, device=None):
    """List of ``[D, H_i*W_i]`` host tensors, one per image (imageretrievalnet.py:358-381)."""
    return _extract_per_image(net, images, image_size, transform, bbxs, ms, print_freq, device, extract_ssl)
