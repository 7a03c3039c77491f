"""``validate`` stage -- ``mdir/stages/validate.py:15-40``: load the network of a
scenario, build its validation tree, run every score under ``torch.no_grad()`` and
return ``({"eval": {metric_key: value}},)``.  Requires a GPU: the score's hot path has
no CPU fallback."""
import numpy as np
import torch

from .events import initialize_processor
from .network import load_network
from .validation import initialize_validation


def validate(params, data, device=None):
    """``device`` defaults to the GPU; it is a parameter only so that the host logic can
    be exercised by the CPU tests with the kernels faked."""
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("mdir_amd.stages.validate needs an MI355X (ROCm) device")
        device = torch.device("cuda")
    np.random.seed(0)
    torch.manual_seed(0)

    assert params.keys() == {"network", "validation", "data"}, params.keys()
    network = load_network(params["network"], device).eval()
    net_defaults = network.network_params.runtime.get("data", {})
    validation = initialize_validation(params["validation"], data=data, params_data=params["data"],
                                       default_criterion=None, net_defaults=net_defaults)
    events = initialize_processor({"progress": {"print_each": 100, "key_suffix": "validation/loss:total"}},
                                  dataroot=None)
    with torch.no_grad():
        for val, valtask in validation.validations(None):
            logger = lambda iteration, size, label, value, dtype, val=val: \
                events.register_data(0, iteration, size, "%s/validation/%s" % (val, label), value, dtype)
            valtask.validate(network, device, logger)
    events.close_epoch()
    return {"eval": {x: y[0] for x, y in events.metadata.metadata().items()}},


def _collate_one(batch):
    """batch_size 1: an image tensor gets its batch axis, an unreadable image stays ``{}``."""
    return batch[0] if isinstance(batch[0], dict) else batch[0].unsqueeze(0)


class EmbeddingOutput:
    """``mdir/components/data/output.py:117-139``: collects one descriptor per image as a
    float64 ``[N,D]`` matrix, NaN rows for unreadable images.  Here the rows are gathered in ONE
    device buffer and copied to the host once, in ``postprocess``."""

    def __init__(self, data, _data_params, *, bbxs=False):
        if not bbxs:
            assert len(data) == 1, len(data)
        self.images, self.bbxs = data if bbxs else (data[0], None)
        self.vecs = None
        self._missing = []

    def preprocess(self):
        return self.images, self.bbxs

    def add(self, index, input_data, output_data):
        if input_data is None and output_data is None:
            self._missing.append(index)
            return
        vec = output_data.reshape(-1)
        if self.vecs is None:
            self.vecs = torch.zeros((len(self.images), vec.numel()), dtype=torch.float32, device=vec.device)
        self.vecs[index].copy_(vec, non_blocking=True)

    def postprocess(self):
        if self.vecs is None:
            return self.images, []
        out = self.vecs.cpu().numpy().astype(np.float64)
        out[self._missing, :] = np.nan
        return self.images, out


OUTPUT_LABELS = {"embedding": EmbeddingOutput}


def infer(params, data, device=None):
    """``infer`` stage -- ``mdir/stages/infer.py:18-64`` for the ``embedding`` output (SURVEY.md
    section 8 row f2): a list of images -> ``(metadata, images, float64 [N,D])``.

    ``params = {"network": {path, runtime}, "data": {"test": {"dataset": {"name": "CirImageList",
    "image_dir", "image_size"}, ["transforms", "mean_std"]}}, "output": {"inference": {"name":
    "embedding", ["bbxs"]}}}``; ``data = (images,)`` or ``(images, bbxs)``."""
    import copy
    import time
    from .datasets import ImagesFromList, initialize_transforms
    from .scenario import path_join
    from .validation import get_dataset_params
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("mdir_amd.stages.infer needs an MI355X (ROCm) device")
        device = torch.device("cuda")
    np.random.seed(0)
    torch.manual_seed(0)

    out_params = copy.deepcopy(params["output"]["inference"])
    out_params.pop("async", None)                      # the single end-of-run copy makes the saver thread moot
    network = load_network(params["network"], device).eval()
    data_params = get_dataset_params(params["data"]["test"], network.network_params.runtime.get("data", {}))
    output = OUTPUT_LABELS[out_params.pop("name")](data, copy.deepcopy(data_params), **out_params)
    images, bbxs = output.preprocess()
    if not images:
        return ({"status": "skipped"},) + output.postprocess()

    ds = copy.deepcopy(data_params["dataset"])
    assert ds.pop("name") == "CirImageList", "only image-list datasets are on the inference path"
    image_dir = ds.pop("image_dir")
    transform = initialize_transforms(data_params["transforms"], data_params["mean_std"])
    # same device-side pieces as extract_vectors_device: uint8 through the loader, one hipGraph
    # replay per input shape, equal-sized images consecutively (rows stay at the caller's indices)
    from . import ops
    from .datasets import ToUint8HWC
    from .graphs import ShapeGraphs, graphs_enabled
    from .networks import ShapeOrder, _Sequential, _gpu_preprocess, batched_loop
    paths = [path_join(image_dir, x) for x in images]
    describe = network
    tail = transform.device_tail() if _gpu_preprocess(device) else None
    import os
    image_size, resize_on_device = ds.pop("image_size"), False
    if tail is not None:
        from .resample import DeviceThumbnail
        resize_on_device = image_size is not None and os.environ.get("MDIR_AMD_GPU_RESIZE", "1") != "0"
        shrink = DeviceThumbnail(image_size) if resize_on_device else (lambda u8: u8)       # the LANCZOS thumbnail on the device
        from .datasets import device_convert
        convert = device_convert(tail)
        transform, describe = ToUint8HWC(), (lambda u8: network(convert(shrink(u8))))
    order = _Sequential(len(paths))
    if graphs_enabled(device):
        describe, order = ShapeGraphs(describe), ShapeOrder(paths, bbxs)
    workers = int(os.environ.get("MDIR_AMD_WORKERS", "8"))
    decode_on_device = tail is not None and (resize_on_device or image_size is None) and workers > 0 \
        and os.environ.get("MDIR_AMD_GPU_JPEG", "1") != "0" and os.environ.get("MDIR_AMD_LOADER", "threads") != "processes"
    dataset = ImagesFromList(root="", images=paths, imsize=image_size, bbxs=bbxs, transform=transform,
                             resize_on_device=resize_on_device, decode_on_device=decode_on_device, **ds)
    from .datasets import make_loader
    loader = make_loader(dataset, order, workers, device, collate_fn=_collate_one)
    t0 = time.time()
    with torch.no_grad():
        batched_loop(loader, order, device, describe, store=lambda i, v: output.add(i, True, v),
                     missing=lambda i: output.add(i, None, None), batches=getattr(network, "supports_batches", False))
    total = time.time() - t0
    metadata = {"stats": {"total_time": int(total), "avg_time": total / len(loader)}}
    return (metadata,) + output.postprocess()


# ---------------------------------------------------------------- whitening stages (mdir/stages/whiten.py:10-87; SURVEY 8 row f3's callers)

class ResourceUsage:
    """``mdir/tools/stats.py:70-131`` as far as the stages report it: current RAM / device memory and the process's cumulative CPU
    and IO counters (``psutil``; the reference's per-process figure from ``nvidia-smi`` has no counterpart here and is None)."""

    def __init__(self):
        self.resources = {}

    def take_current_stats(self):
        import psutil
        self.resources["ram_memory_gib"] = round(psutil.Process().memory_info().vms / 2 ** 30, 3)
        if torch.cuda.is_available():
            self.resources["gpu"] = {"memory_nvidia_gib": None, "memory_torch_gib": round(torch.cuda.memory_allocated() / 2 ** 30, 3)}
        return self

    def get_resources(self):
        import time
        import psutil
        proc = psutil.Process()
        with proc.oneshot():
            cpu = proc.cpu_times()
            stats = {"cpu": {"user_s": int(cpu.user), "system_s": int(cpu.system), "children_user_s": int(cpu.children_user),
                             "children_system_s": int(cpu.children_system), "proc_wall_s": int(time.time() - proc.create_time())}}
            stats["cpu"]["tree_used_s"] = sum(stats["cpu"][k] for k in ("user_s", "system_s", "children_user_s", "children_system_s"))
            stats["cpu"]["avg_cores"] = round(stats["cpu"]["tree_used_s"] / max(stats["cpu"]["proc_wall_s"], 1), 1)
            io = proc.io_counters()
            stats["io"] = {"read_count": io.read_count, "write_count": io.write_count,
                           "read_gib": round(io.read_bytes / 2 ** 30, 3), "write_gib": round(io.write_bytes / 2 ** 30, 3)}
        return {**self.resources, **stats}


def whiten(params, data, device="cuda"):
    """Apply a pre-computed whitening to ``[N,D]`` descriptors (``stages/whiten.py:10-24``): ``(metadata, names, [N,d])``."""
    import time
    from .whiten import whitenapply
    dimensions = params.pop("dimensions", None) or None
    assert not params, params.keys()
    whitening, names, values = data
    assert len(names) == len(values)
    resources = ResourceUsage()
    time0 = time.time()
    whitened = whitenapply(values.T, whitening["m"], whitening["P"], dimensions, device=device)
    metadata = {"timings": {"whitening_apply": round(time.time() - time0, 2)},
                "resource_usage": resources.take_current_stats().get_resources()}
    return metadata, names, whitened.T


def learn_lw_whitening(params, data, device="cuda"):
    """Learn the supervised whitening from (query, positive) NAME pairs (``stages/whiten.py:27-69``): ``(metadata, {'m','P'})``.
    A covariance that is not positive definite is retried on shrinking random subsets of the pairs, up to 100 trials, as upstream."""
    import sys
    import time
    from .whiten import whitenlearn
    assert not params
    names, values, queries, positives = data
    assert len(names) == len(values)
    assert len(queries) == len(positives)
    values = values.astype(np.float64).T
    name_index = {x: i for i, x in enumerate(names)}
    qidxs = np.array([name_index[x] for x in queries])
    pidxs = np.array([name_index[x] for x in positives])
    resources = ResourceUsage()
    time0 = time.time()
    max_trials, max_excluded, trial = 100, 0.95, 0
    while True:
        try:
            if trial == 0:
                qwhit, pwhit = qidxs, pidxs
            else:
                idxs = np.random.permutation(len(qidxs))[:int(len(qidxs) * (1 - trial / max_trials * max_excluded))]
                print("Using subset of queries (%s/%s) trial %s" % (len(idxs), len(qidxs), trial), file=sys.stderr)
                qwhit, pwhit = qidxs[idxs], pidxs[idxs]
            whit_m, whit_p = whitenlearn(values, qwhit, pwhit, device=device)
            break
        except np.linalg.LinAlgError as e:
            if str(e) != "Matrix is not positive definite" or trial >= max_trials - 1:
                raise
            trial += 1
    metadata = {"stats": {"failed_times": trial, "vectors_used": round(len(qwhit) / float(len(qidxs)), 2), "vectors_total": len(qidxs)},
                "timings": {"whitening_learn": round(time.time() - time0, 2)},
                "resource_usage": resources.take_current_stats().get_resources()}
    return metadata, {"m": whit_m, "P": whit_p}


def learn_pca_whitening(params, data, device="cuda"):
    """Learn the PCA whitening of ``[N,D]`` descriptors (``stages/whiten.py:72-87``): ``(metadata, {'m','P'})``."""
    import time
    from .whiten import pcawhitenlearn
    shrink = params.pop("shrink", None) or None
    assert not params
    values, = data
    values = values.astype(np.float64).T
    resources = ResourceUsage()
    time0 = time.time()
    whit_m, whit_p = pcawhitenlearn(values, shrink, device=device)
    metadata = {"timings": {"whitening_learn": round(time.time() - time0, 2)},
                "resource_usage": resources.take_current_stats().get_resources()}
    return metadata, {"m": whit_m, "P": whit_p}


def paste_pca_normalize(params, data, device="cuda"):
    """Concatenate descriptor matrices side by side, optionally keep the ``dimensions`` principal directions, L2-normalise the rows
    (``stages/whiten.py:90-118``): ``(metadata, [N, sum D])``.  The PCA is upstream's: the SCALAR mean of the whole matrix is
    subtracted, the eigenvectors of ``value.T @ value`` with the largest eigenvalues span the subspace the rows are projected onto
    (the dimension of the rows does not change).  The Gram matrix and the projection run on the f64 matrix cores, the symmetric
    eigen-decomposition next to the data (upstream: ``np.linalg.eig`` on the host)."""
    import time
    from . import ops
    from .whiten import _as_f64
    dimensions = params.pop("dimensions") or None
    assert not params
    assert len(set(len(x) for x in data)) == 1
    if data[0].shape == (0,):
        return {}, data[0]
    value = np.concatenate(data, axis=1)
    in_dtype = value.dtype
    if dimensions:
        resources = ResourceUsage()
        time0 = time.time()
        value = value - np.mean(value)
        vt = _as_f64(value.T, device)                                    # [D, N]
        eigval, eigvec = torch.linalg.eigh(ops.gram_f64(vt))             # ascending: the last `dimensions` columns are the largest
        vecs = eigvec[:, -dimensions:].contiguous()
        proj = ops.project_f64(ops.gram_f64(vecs), vt)                   # (V V^T) value^T  -> [D, N]
        value = proj.t().contiguous().cpu().numpy()
        if in_dtype in (np.float32, np.float64):
            value = value.astype(in_dtype, copy=False)                   # upstream's statements run in the dtype of their input
        metadata = {"timings": {"pca_compute": round(time.time() - time0, 2)},
                    "resource_usage": resources.take_current_stats().get_resources()}
    else:
        metadata = {}
    value = value / np.expand_dims(np.linalg.norm(value, axis=1), axis=1)
    return metadata, value
