"""Convolutional feature stacks declared with torch only (no torchvision, no downloads).

``init_network`` of the reference takes ``torchvision.models.<arch>`` and keeps the
convolutional children (``cirtorch/networks/imageretrievalnet.py:155-181``).  The
module trees below reproduce torchvision's child order and parameter names, so a
reference ``state_dict`` (``features.0.weight``, ``features.4.0.conv1.weight`` ...)
loads unchanged.  The convolutions themselves stay on PyTorch-ROCm / MIOpen: the
backbone is not part of the hand-written hot path (BASELINE.json north_star).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride, 1, bias=False)


def _bn_act(x, bn, residual=None, relu=True):
    """``relu(bn(x) + residual)`` on a fresh convolution output.  At inference on the GPU the three
    full-tensor passes are one in-place kernel (``mdx_bn_act``: a third of the trunk's GPU time and
    400 of its 1100 launches on ResNet101); training / CPU tensors / autograd keep the module calls."""
    if (x.is_cuda and not bn.training and not torch.is_grad_enabled() and bn.track_running_stats
            and x.dtype == torch.float32 and _fused_trunk()):
        from . import ops
        res = residual.contiguous() if residual is not None else None
        return ops.bn_act_(x.contiguous(), bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, res, relu)
    out = bn(x)
    if residual is not None:
        out = out + residual
    return F.relu(out, inplace=True) if relu else out


def _fused_trunk():
    return os.environ.get("MDIR_AMD_FUSED_TRUNK", "1") != "0"


def _own_conv1x1(conv, residual):
    """Which stride-1 1x1 convolutions run on libmdx's GEMM with the epilogue fused (``mdx_conv1x1_bn_act``) instead of MIOpen +
    ``mdx_bn_act``.  ``MDIR_AMD_CONV1X1``: ``auto`` (default) = where it is faster, by a FIXED rule (no run-time timing: results
    do not depend on a measurement): the expand convolutions (the ones that add the identity: K is short and the epilogue pass
    they save costs as much as the GEMM) and convolutions with <= 64 output channels (MIOpen's choice there runs at 40 TFLOP/s);
    the long-K reduce convolutions stay with the library GEMM, which runs at 110-135 TFLOP/s against 95-100 here
    (``profiles/r03_conv1x1.md``).  ``1`` = every supported one, ``0`` = none."""
    mode = os.environ.get("MDIR_AMD_CONV1X1", "auto")
    if mode == "0":
        return False
    return mode == "1" or residual is not None or conv.out_channels <= 64


def _conv_bn_act(conv, bn, x, residual=None, relu=True):
    """``relu(bn(conv(x)) + residual)`` of a Bottleneck's 1x1 convolutions: one kernel where ``_own_conv1x1`` says so (the
    convolution output is written once, finished), else the library convolution followed by ``mdx_bn_act``."""
    if (x.is_cuda and x.dtype == torch.float32 and not bn.training and not torch.is_grad_enabled() and bn.track_running_stats
            and _fused_trunk() and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0)
            and conv.groups == 1 and conv.bias is None and _own_conv1x1(conv, residual)):
        from . import ops
        if ops.conv1x1_supported(conv.in_channels, conv.out_channels):
            # the transposed copy of the weights is cached on the module, keyed by WHICH tensor the parameter is (storage
            # pointer, shape) and by its in-place version: `load_state_dict(assign=True)`, `conv.weight = Parameter(...)`,
            # `conv.weight.data = ...` and `.double().float()` replace the tensor without bumping `_version`
            w = conv.weight
            key = (w.data_ptr(), tuple(w.shape), w._version, x.device)
            wt = getattr(conv, "_mdx_wt", None)
            if wt is None or wt[0] != key:
                wt = (key, ops.conv1x1_transpose_weights(w.detach().contiguous()))
                conv._mdx_wt = wt
            res = residual.contiguous() if residual is not None else None
            return ops.conv1x1_bn_act(x.contiguous(), wt[1], bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, res, relu)
    return _bn_act(conv(x), bn, residual, relu)


def _downsample(mod, x):
    if isinstance(mod, nn.Sequential) and len(mod) == 2 and isinstance(mod[1], nn.BatchNorm2d):
        return _bn_act(mod[0](x), mod[1], None, relu=False)
    return mod(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(cin, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else _downsample(self.downsample, x)
        out = _bn_act(self.conv1(x), self.bn1)
        return _bn_act(self.conv2(out), self.bn2, idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else _downsample(self.downsample, x)
        out = _conv_bn_act(self.conv1, self.bn1, x)
        out = _bn_act(self.conv2(out), self.bn2)
        return _conv_bn_act(self.conv3, self.bn3, out, idt)


class TrunkSequential(nn.Sequential):
    """``nn.Sequential`` (same child names, same state dict) whose inference pass on the GPU folds
    the elementwise modules that follow a convolution into one in-place kernel:
    ``Conv2d(bias=False) -> BatchNorm2d [-> ReLU]`` (ResNet stem) and ``Conv2d(bias) -> ReLU``
    (every VGG / AlexNet layer: the bias add and the ReLU are two full-tensor passes otherwise)."""

    def forward(self, x):
        if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and _fused_trunk()):
            return super().forward(x)
        from . import ops
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, nn.Conv2d) and m.padding_mode == "zeros" and not isinstance(m.padding, str):
                if (m.bias is None and isinstance(nxt, nn.BatchNorm2d) and not nxt.training and nxt.track_running_stats):
                    relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                    x = ops.bn_act_(m(x).contiguous(), nxt.running_mean, nxt.running_var, nxt.weight, nxt.bias, nxt.eps,
                                    None, relu)
                    i += 3 if relu else 2
                    continue
                if m.bias is not None and isinstance(nxt, nn.ReLU):
                    y = F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)
                    x = ops.bn_act_(y.contiguous(), None, None, None, m.bias, 0.0, None, True)
                    i += 2
                    continue
            x = m(x)
            i += 1
        return x


_RESNET = {"resnet18": (BasicBlock, (2, 2, 2, 2)), "resnet34": (BasicBlock, (3, 4, 6, 3)),
           "resnet50": (Bottleneck, (3, 4, 6, 3)), "resnet101": (Bottleneck, (3, 4, 23, 3)),
           "resnet152": (Bottleneck, (3, 8, 36, 3))}


def _resnet_features(arch):
    block, layers = _RESNET[arch]
    state = {"cin": 64}

    def make_layer(planes, blocks, stride):
        down = None
        if stride != 1 or state["cin"] != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(state["cin"], planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        mods = [block(state["cin"], planes, stride, down)]
        state["cin"] = planes * block.expansion
        mods += [block(state["cin"], planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    # children of torchvision's ResNet minus (avgpool, fc): imageretrievalnet.py:172-173
    return [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
            nn.MaxPool2d(3, 2, 1), make_layer(64, layers[0], 1), make_layer(128, layers[1], 2),
            make_layer(256, layers[2], 2), make_layer(512, layers[3], 2)]


_VGG = {"vgg11": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
        "vgg13": [64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
        "vgg16": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
        "vgg19": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M",
                  512, 512, 512, 512, "M"]}


def _vgg_features(arch):
    mods, cin = [], 3
    for v in _VGG[arch]:
        if v == "M":
            mods.append(nn.MaxPool2d(2, 2))
        else:
            mods += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return mods[:-1]   # the last max-pool is dropped: imageretrievalnet.py:170-171


def _alexnet_features():
    mods = [nn.Conv2d(3, 64, 11, 4, 2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(64, 192, 5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2),
            nn.Conv2d(192, 384, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(384, 256, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, 3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2)]
    return mods[:-1]   # imageretrievalnet.py:168-169


# ---- DenseNet / SqueezeNet (imageretrievalnet.py:73-78, 175-180; round 6).  torchvision's module names, so that a reference
# state_dict loads unchanged: features.4.denselayer1.norm1.weight, features.5.conv.weight, features.3.squeeze.weight ...
class _DenseLayer(nn.Module):
    def __init__(self, cin, growth, bn_size):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv1 = nn.Conv2d(cin, bn_size * growth, 1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(bn_size * growth, growth, 3, padding=1, bias=False)

    def forward(self, feats):
        x = torch.cat(feats, 1) if isinstance(feats, (list, tuple)) else feats
        x = self.conv1(self.relu1(self.norm1(x)))
        return self.conv2(self.relu2(self.norm2(x)))


class _DenseBlock(nn.ModuleDict):
    def __init__(self, layers, cin, growth, bn_size):
        super().__init__()
        for i in range(layers):
            self.add_module("denselayer%d" % (i + 1), _DenseLayer(cin + i * growth, growth, bn_size))

    def forward(self, x):
        feats = [x]
        for layer in self.values():
            feats.append(layer(feats))
        return torch.cat(feats, 1)


class _Transition(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module("norm", nn.BatchNorm2d(cin))
        self.add_module("relu", nn.ReLU(inplace=True))
        self.add_module("conv", nn.Conv2d(cin, cout, 1, bias=False))
        self.add_module("pool", nn.AvgPool2d(2, 2))


_DENSENET = {"densenet121": (32, (6, 12, 24, 16), 64), "densenet169": (32, (6, 12, 32, 32), 64),
             "densenet201": (32, (6, 12, 48, 32), 64), "densenet161": (48, (6, 12, 36, 24), 96)}


def _densenet_features(arch):
    growth, blocks, c = _DENSENET[arch]
    mods = [nn.Conv2d(3, c, 7, 2, 3, bias=False), nn.BatchNorm2d(c), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)]
    for i, layers in enumerate(blocks):
        mods.append(_DenseBlock(layers, c, growth, 4))
        c += layers * growth
        if i != len(blocks) - 1:
            mods.append(_Transition(c, c // 2))
            c //= 2
    # children of torchvision's `features` (... norm5) + a ReLU: imageretrievalnet.py:175-177
    return mods + [nn.BatchNorm2d(c), nn.ReLU(inplace=True)]


class _Fire(nn.Module):
    def __init__(self, cin, squeeze, e1, e3):
        super().__init__()
        self.squeeze = nn.Conv2d(cin, squeeze, 1)
        self.squeeze_activation = nn.ReLU(inplace=True)
        self.expand1x1 = nn.Conv2d(squeeze, e1, 1)
        self.expand1x1_activation = nn.ReLU(inplace=True)
        self.expand3x3 = nn.Conv2d(squeeze, e3, 3, padding=1)
        self.expand3x3_activation = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.squeeze_activation(self.squeeze(x))
        return torch.cat([self.expand1x1_activation(self.expand1x1(x)), self.expand3x3_activation(self.expand3x3(x))], 1)


def _squeezenet_features(arch):
    pool = lambda: nn.MaxPool2d(3, 2, ceil_mode=True)
    if arch == "squeezenet1_0":
        return [nn.Conv2d(3, 96, 7, 2), nn.ReLU(inplace=True), pool(), _Fire(96, 16, 64, 64), _Fire(128, 16, 64, 64), _Fire(128, 32, 128, 128),
                pool(), _Fire(256, 32, 128, 128), _Fire(256, 48, 192, 192), _Fire(384, 48, 192, 192), _Fire(384, 64, 256, 256), pool(),
                _Fire(512, 64, 256, 256)]
    # all children of torchvision's `features`: imageretrievalnet.py:178-179
    return [nn.Conv2d(3, 64, 3, 2), nn.ReLU(inplace=True), pool(), _Fire(64, 16, 64, 64), _Fire(128, 16, 64, 64), pool(),
            _Fire(128, 32, 128, 128), _Fire(256, 32, 128, 128), pool(), _Fire(256, 48, 192, 192), _Fire(384, 48, 192, 192),
            _Fire(384, 64, 256, 256), _Fire(512, 64, 256, 256)]


# imageretrievalnet.py:58-78
OUTPUT_DIM = {"alexnet": 256, "vgg11": 512, "vgg13": 512, "vgg16": 512, "vgg19": 512, "resnet18": 512,
              "resnet34": 512, "resnet50": 2048, "resnet101": 2048, "resnet152": 2048,
              "densenet121": 1024, "densenet169": 1664, "densenet201": 1920, "densenet161": 2208,
              "squeezenet1_0": 512, "squeezenet1_1": 512}


def build_features(architecture):
    """List of feature modules for ``architecture`` (randomly initialised)."""
    if architecture == "alexnet":
        return _alexnet_features()
    if architecture in _VGG:
        return _vgg_features(architecture)
    if architecture in _RESNET:
        return _resnet_features(architecture)
    if architecture in _DENSENET:
        return _densenet_features(architecture)
    if architecture in ("squeezenet1_0", "squeezenet1_1"):
        return _squeezenet_features(architecture)
    raise ValueError("Unsupported or unknown architecture: {}!".format(architecture))
