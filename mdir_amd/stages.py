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
