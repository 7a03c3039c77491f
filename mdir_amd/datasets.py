"""Image lists, test-set configuration and the ``pil2np | totensor | normalize`` chain.

CPU-side loader of the hot path; mirrors
``cirtorch/datasets/genericdataset.py:10-81`` (ImagesFromList),
``datahelpers.py:24-50`` (pil_loader, imresize), ``testdataset.py:4-38``
(configdataset), ``utils/general.py:4-11`` (get_data_root) and mdir's transform DSL
(``mdir/components/data/transform/__init__.py:35-44``, ``core_transforms.py:33-63``).
Only the transforms eval.py uses are provided; CLAHE & co. are CPU/OpenCV
preprocessing upstream of the tensor this path consumes.
"""
import os
import pickle
import sys

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image, ImageFile

ImageFile.LOAD_TRUNCATED_IMAGES = True   # datahelpers.py:7

DATASETS = ["oxford5k", "paris6k", "roxford5k", "rparis6k", "247tokyo1k"]


def get_root():
    if os.environ.get("CIRTORCH_ROOT", ""):
        return os.environ["CIRTORCH_ROOT"]
    return os.path.abspath(os.path.join(os.path.dirname(os.path.realpath(__file__)), ".."))


def get_data_root():
    return os.path.join(get_root(), "data")


def pil_loader(path):
    """RGB PIL image, or the OSError instance if it cannot be opened (reference contract)."""
    try:
        with open(path, "rb") as f:
            return Image.open(f).convert("RGB")
    except OSError as e:
        return e


default_loader = pil_loader


def imresize(img, imsize):
    """Aspect-preserving DOWN-scale so that the longer side is <= imsize
    (``thumbnail`` + ANTIALIAS, which Pillow >= 10 calls LANCZOS)."""
    img.thumbnail((imsize, imsize), Image.LANCZOS)
    return img


class ImagesFromList(data.Dataset):
    """``resize_on_device=True`` (not in the reference): the down-scaling to ``imsize`` is left to the consumer
    (``mdir_amd.resample.DeviceThumbnail``, same pixels) for the images the device path covers; the others are
    shrunk here as always.  ``decode_on_device=True`` (with it): a baseline JPEG file is only entropy-decoded here and
    handed over as ``mdir_amd.jpeg.JpegCoefficients`` -- IDCT, upsampling and colour conversion happen on the device
    (``mdir_amd.jpeg.pixels``, same pixels); other files, and boxes that stick out of the image, take the usual route."""

    def __init__(self, root, images, imsize=None, bbxs=None, transform=None, loader=default_loader,
                 ignore_errors=False, resize_on_device=False, decode_on_device=False):
        images_fn = [os.path.join(root, images[i]) for i in range(len(images))]
        if len(images_fn) == 0:
            raise RuntimeError("Dataset contains 0 images!")
        self.root, self.images, self.imsize, self.images_fn = root, images, imsize, images_fn
        self.bbxs, self.transform, self.loader, self.ignore_errors = bbxs, transform, loader, ignore_errors
        self.resize_on_device = resize_on_device
        # items decoded on the device leave this object as JPEG coefficients: no crop, no resize and NO transform is
        # applied to them here, so the only transform that may be configured is the one the device tail replaces
        if decode_on_device and not (transform is None or isinstance(transform, ToUint8HWC)):
            raise ValueError("decode_on_device=True hands JPEG coefficients to the device pipeline and cannot apply "
                             "transform %r to them: use ToUint8HWC (or None)" % (transform,))
        self.decode_on_device = decode_on_device and (resize_on_device or imsize is None) and loader is default_loader

    def _coefficients(self, index):
        """The file as JPEG coefficients for the device, or None (not such a file, or a case for the host route).

        Never raises and never decides an error: whatever is wrong with a file -- unreadable, not an image, a header Pillow
        refuses, a stream the entropy decoder gives up on -- sends the item down the host route, where ``self.loader``
        reports it exactly as the reference does (genericdataset.py:52-59: the OSError re-raised, or ``{}`` under
        ``ignore_errors``).  Pillow parses the header FIRST (``Image.open`` reads no pixel data): the library's own parser
        only ever sees files Pillow has accepted as JPEG of a sane size, on top of its own checks (bounds-checked and fuzzed
        under ASan/UBSan, tests/test_fuzz_asan.py)."""
        from . import jpeg
        from .resample import on_device
        try:
            with open(self.images_fn[index], "rb") as f:
                data = f.read()
            if data[:2] != b"\xff\xd8":
                return None
            import io
            with Image.open(io.BytesIO(data)) as seen:      # raises for what Pillow does not recognise, and for decompression bombs
                if seen.format != "JPEG" or seen.mode not in ("RGB", "L"):
                    return None
                seen_size = seen.size
            # beyond Pillow's MAX_IMAGE_PIXELS it warns about the picture: let it (entropy_decode declines, the host route decodes)
            item = jpeg.entropy_decode(data, self.bbxs[index] if self.bbxs else None, max_pixels=Image.MAX_IMAGE_PIXELS or jpeg.MAX_PIXELS)
            if item is None or item.size != seen_size:
                return None
        except Exception:
            return None
        w, h = item.size
        if item.box:
            if not jpeg.box_on_device(item.box, w, h):
                return None
            w, h = int(item.box[2] - item.box[0]), int(item.box[3] - item.box[1])
        # the thumbnail must be one the device makes (or none at all): otherwise Pillow shrinks it here, from its own decode
        if self.imsize is not None and on_device(w, h, self.imsize) is None:
            from .resample import thumbnail_size
            if thumbnail_size(w, h, self.imsize) not in (None, (w, h)):
                return None
        return item

    def __getitem__(self, index):
        path = self.images_fn[index]
        if self.decode_on_device:
            item = self._coefficients(index)
            if item is not None:
                return item
        img = self.loader(path)
        if isinstance(img, Exception):
            sys.stderr.write("Warning: Image '%s' was not found\n" % path)
            if self.ignore_errors:
                return {}
            raise img
        if self.bbxs and self.bbxs[index]:
            img = img.crop(self.bbxs[index])
        if self.imsize is not None:
            deferred = False
            if self.resize_on_device:
                from .resample import on_device
                deferred = on_device(img.size[0], img.size[1], self.imsize) is not None
            if not deferred:
                img = imresize(img, self.imsize)
        if self.transform is not None:
            img = self.transform(img)
        return img

    def __len__(self):
        return len(self.images_fn)


class ThreadedLoader:
    """``DataLoader(dataset, batch_size=1, sampler=order, num_workers=n, pin_memory=True)`` with THREADS instead of worker
    processes, for datasets whose items are made in C with the GIL released (Pillow's JPEG decode and resize, numpy
    copies): no fork, no pickling, no shared-memory files -- a decoded image goes from the decoding thread straight into
    pinned memory.  Yields items in the sampler's order (a batch dimension of 1 in front, like the DataLoader); at most
    ``2 * workers`` items are in flight.  An item that raises is re-raised at its turn."""

    def __init__(self, dataset, sampler, workers, pin_memory=True):
        self.dataset, self.sampler, self.workers, self.pin = dataset, sampler, max(1, int(workers)), pin_memory

    def __len__(self):
        return len(self.sampler)

    def _load(self, index):
        item = self.dataset[index]
        if isinstance(item, torch.Tensor):
            item = item.unsqueeze(0)
            if self.pin and torch.cuda.is_available():
                item = item.pin_memory()
        elif self.pin and hasattr(item, "pin_memory") and torch.cuda.is_available():
            item = item.pin_memory()                        # mdir_amd.jpeg.JpegCoefficients
        return item

    def __iter__(self):
        import collections
        from concurrent.futures import ThreadPoolExecutor
        pending = collections.deque()
        with ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="mdir-loader") as pool:
            try:
                for index in self.sampler:
                    pending.append(pool.submit(self._load, index))
                    if len(pending) >= 2 * self.workers:
                        yield pending.popleft().result()
                while pending:
                    yield pending.popleft().result()
            finally:
                for f in pending:
                    f.cancel()


def make_loader(dataset, sampler, workers, device, collate_fn=None):
    """The batch-size-1 loader of the extraction loops.  Threads by default: with the images decoded in worker PROCESSES
    (``MDIR_AMD_LOADER=processes``: torch's DataLoader, the reference's choice, imageretrievalnet.py:284-287) the main
    process's launches slow down threefold as soon as three workers are busy -- 95 descriptors/s instead of 203 on a
    16-size JPEG list with ResNet101 (tools/list_workers_probe.py) -- while Pillow's decoder releases the GIL."""
    if workers > 0 and os.environ.get("MDIR_AMD_LOADER", "threads") != "processes":
        return ThreadedLoader(dataset, sampler, workers, pin_memory=torch.device(device).type == "cuda")
    kw = {"collate_fn": collate_fn} if collate_fn is not None else {}
    return data.DataLoader(dataset, batch_size=1, shuffle=False, sampler=sampler, num_workers=workers, pin_memory=True, **kw)


def config_imname(cfg, i):
    return os.path.join(cfg["dir_images"], cfg["imlist"][i] + cfg["ext"])


def config_qimname(cfg, i):
    return os.path.join(cfg["dir_images"], cfg["qimlist"][i] + cfg["qext"])


def configdataset(dataset, dir_main):
    dataset = dataset.lower()
    if dataset not in DATASETS:
        raise ValueError("Unknown dataset: {}!".format(dataset))
    gnd_fname = os.path.join(dir_main, dataset, "gnd_{}.pkl".format(dataset))
    with open(gnd_fname, "rb") as f:
        cfg = pickle.load(f)
    cfg["gnd_fname"] = gnd_fname
    cfg["ext"] = cfg["qext"] = ".jpg"
    cfg["dir_data"] = os.path.join(dir_main, dataset)
    cfg["dir_images"] = os.path.join(cfg["dir_data"], "jpg")
    cfg["n"], cfg["nq"] = len(cfg["imlist"]), len(cfg["qimlist"])
    cfg["im_fname"], cfg["qim_fname"] = config_imname, config_qimname
    cfg["dataset"] = dataset
    return cfg


# ------------------------------------------------------------------ transforms

class Pil2Numpy:
    """PIL -> float32 HWC array in [0,1] (core_transforms.py:56-60)."""

    def __call__(self, *pics):
        return [np.array(x.convert("RGB"), dtype=np.float32) / 255.0 for x in pics]


class ToTensor:
    """HWC array (or PIL) -> CHW float tensor; uint8 inputs are scaled by 1/255 like
    torchvision's ToTensor (core_transforms.py:33-36)."""

    def __call__(self, *pics):
        out = []
        for x in pics:
            if isinstance(x, Image.Image):
                x = np.array(x)
            t = torch.from_numpy(np.ascontiguousarray(x if x.ndim == 3 else x[:, :, None]).transpose(2, 0, 1))
            out.append(t.float().div(255) if t.dtype == torch.uint8 else t.float())
        return out


class Normalize:
    def __init__(self, mean, std, strict_shape=True):
        if isinstance(strict_shape, str):
            strict_shape = strict_shape.lower() != "false"
        assert len(mean) == len(std)
        self.mean, self.std, self.strict_shape = list(mean), list(std), bool(strict_shape)

    def __call__(self, *pics):
        out = []
        for pic in pics:
            c = pic.size(0)
            if self.strict_shape:
                assert c == len(self.mean), (c, len(self.mean))
            else:
                assert c <= len(self.mean), (c, len(self.mean))
            mean = torch.tensor(self.mean[:c], dtype=pic.dtype).view(-1, 1, 1)
            std = torch.tensor(self.std[:c], dtype=pic.dtype).view(-1, 1, 1)
            out.append((pic - mean) / std)
        return out


class ApplyClahe:
    """``apply_clahe[:clip_limit[:colorspace[:grid_size]]]`` of the scenario DSL (``photometric_transforms.ApplyClahe``,
    photometric_transforms.py:28-36; defaults 4 / lab / 8): CLAHE on the lightness of the image.  The reference runs it in
    the loader workers through OpenCV; here it is device work (``mdx_clahe_u8_to_chw``, fused with ``totensor | normalize``,
    see :meth:`Compose.device_tail`), so this object only carries the parameters: calling it on the host -- a transform
    chain the device tail does not cover, or ``MDIR_AMD_GPU_PREPROCESS=0`` -- raises (no CPU fallback, and no OpenCV)."""

    def __init__(self, clip_limit=4, colorspace="lab", grid_size=8):
        self.clip_limit = int(clip_limit)
        self.colorspace = str(colorspace).lower()
        if self.colorspace != "lab":
            raise NotImplementedError("apply_clahe: colorspace %r (only 'lab', the one the iccv19 scenarios use)" % colorspace)
        self.grid_size = tuple(int(g) for g in grid_size) if isinstance(grid_size, (tuple, list)) else (int(grid_size), int(grid_size))

    def __call__(self, *pics):
        raise RuntimeError("apply_clahe runs on the MI355X only (chain `pil2np | apply_clahe | totensor | normalize` with "
                           "MDIR_AMD_GPU_PREPROCESS on): there is no CPU fallback")


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, *pics):
        for t in self.transforms:
            pics = t(*pics)
        return pics[0] if len(pics) == 1 else pics

    def device_tail(self):
        """``(mean, std)`` when the whole chain is the PIL -> normalised CHW tensor conversion
        (``pil2np | totensor | normalize`` of the scenarios, or cirtorch's ``ToTensor, Normalize``):
        extraction then ships uint8 pixels and does this arithmetic on the GPU (``mdx_u8_to_chw``,
        same fp32 operation order).  ``None`` for any other chain."""
        t = list(self.transforms)
        clahe = None
        if len(t) == 4 and isinstance(t[0], Pil2Numpy) and isinstance(t[1], ApplyClahe):
            # the CLAHE networks' chain: a third element carries the CLAHE parameters (mdx_clahe_u8_to_chw does all of it)
            clahe, t = {"clip_limit": t[1].clip_limit, "grid": t[1].grid_size}, t[2:]
        elif len(t) == 3 and isinstance(t[0], Pil2Numpy):
            t = t[1:]
        if len(t) == 2 and isinstance(t[0], ToTensor) and isinstance(t[1], Normalize) and len(t[1].mean) == 3 \
                and t[1].strict_shape:
            return (list(t[1].mean), list(t[1].std), clahe) if clahe else (list(t[1].mean), list(t[1].std))
        return None


class ToUint8HWC:
    """Loader-side half of the split conversion: PIL -> uint8 ``[H,W,3]`` tensor (no arithmetic)."""

    def __call__(self, pic):
        return torch.from_numpy(np.array(pic.convert("RGB")))


TRANSFORMS = {"totensor": ToTensor, "normalize": Normalize, "pil2np": Pil2Numpy, "apply_clahe": ApplyClahe}


def device_convert(tail):
    """The device half of a transform chain ``Compose.device_tail()`` recognised: uint8 ``[B,H,W,3]`` -> normalised fp32
    ``[B,3,H,W]`` (``mdx_u8_to_chw``, or ``mdx_clahe_u8_to_chw`` for the CLAHE networks' chain)."""
    from . import ops
    if len(tail) > 2 and tail[2]:
        return lambda u8: ops.clahe_u8_to_chw(u8, tail[2]["clip_limit"], tail[2]["grid"], tail[0], tail[1])
    return lambda u8: ops.u8_to_chw(u8, tail[0], tail[1])


def initialize_transforms(augmentations, mean_std):
    """Parse ``"a | b:arg:arg"``; ``normalize`` receives ``mean_std`` first
    (transform/__init__.py:35-44)."""
    trans = []
    for aug in [x.strip() for x in augmentations.split("|") if x.strip()]:
        tname, *args = aug.split(":", 1)
        args = args[0].split(":") if args else []
        if tname not in TRANSFORMS:
            raise KeyError("transform '%s' is outside the MI355X hot path (only %s are provided)"
                           % (tname, sorted(TRANSFORMS)))
        trans.append(TRANSFORMS[tname](*(list(mean_std) + args)) if "normalize" in aug else TRANSFORMS[tname](*args))
    return Compose(trans)
