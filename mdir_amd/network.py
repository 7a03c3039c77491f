"""The network object the evaluation code drives -- the slice of
``mdir/learning/network.py`` that inference needs.

``extract_vectors`` receives this wrapper, not an ``nn.Module``
(``cirscore.py:54`` -> ``network.py:88-89``): it must offer ``.eval()``,
``.meta['out_channels']``, ``.model`` and ``__call__(image) = wrappers[stage](image,
model)``.  ``SingleNetwork`` / ``CirNetwork`` keep the reference's constructor,
``overlay_params`` and checkpoint layout (``{"type","frozen","network_params":{"model",
"runtime"},"model_state"}``, network.py:142-170).  Training-side methods
(parameters, freeze, graphs) and ``SequentialNetwork`` are out of scope.
"""
import copy
from collections import namedtuple

import torch

from .networks import init_network
from .wrapper import initialize_wrappers


def init_cirnet(**params):
    """``mdir/components/model/network/cirnet.py:10-22`` minus the model-zoo download."""
    for key in ["local_whitening", "pooling", "regional", "whitening", "pretrained"]:
        if key not in params:
            raise ValueError("Key '%s' not in params" % key)
    params["mean"] = [0.485, 0.456, 0.406]
    params["std"] = [0.229, 0.224, 0.225]
    params["architecture"] = params.pop("cir_architecture")
    net = init_network(params)
    net.meta["in_channels"] = 3
    net.meta["out_channels"] = net.meta["outputdim"]
    return net


MODEL_LABELS = {"cirnet": init_cirnet}


def initialize_model(params):
    return MODEL_LABELS[params.pop("architecture")](**params)


class SingleNetwork:
    TRAIN, EVAL = "train", "eval"
    NetworkParams = namedtuple("NetworkParams", ["model", "runtime"])

    def __init__(self, model, network_params, device, frozen):
        self.meta = {"in_channels": model.meta["in_channels"], "out_channels": model.meta["out_channels"]}
        self.network_params = network_params
        wrappers = network_params.runtime.get("wrappers", "")
        if isinstance(wrappers, dict):
            assert wrappers.keys() == {"train", "eval"}, wrappers.keys()
            self.wrappers = {x: initialize_wrappers(wrappers[x], device) for x in wrappers}
        else:
            self.wrappers = {x: initialize_wrappers(wrappers, device) for x in ["train", "eval"]}
        self.frozen = network_params.runtime.get("frozen", False) or frozen
        self.model = model.to(device)
        self.stage = None
        if self.frozen:
            self.eval()
        extra = network_params.runtime.keys() - {"data", "wrappers", "frozen"}
        assert not extra, extra
        extra = network_params.runtime.get("data", {}).keys() - {"mean_std", "transforms"}
        assert not extra, extra

    supports_batches = True          # the wrapper chain answers one row per image for a batch (wrapper.Compose)

    def __call__(self, image):
        return self.wrappers[self.stage](image, self.model)

    def eval(self):
        self.model.eval()
        self.stage = self.EVAL
        return self

    def overlay_params(self, new_params, device):
        if not new_params:
            return self
        new_params["runtime"]["frozen"] = True
        network_params = self.NetworkParams(self.network_params.model, new_params.pop("runtime"))
        assert not new_params
        return self.__class__(self.model, network_params, device, frozen=True)

    def state_dict(self):
        return {"net": {"type": self.__class__.__name__, "frozen": self.frozen,
                        "network_params": self.network_params._asdict(),
                        "model_state": self.model.state_dict()}}

    @classmethod
    def initialize_from_state(cls, state_dict, device, params, runtime):
        assert state_dict.keys() == {"net"}, state_dict.keys()
        checkpoint = state_dict["net"]
        assert checkpoint.keys() == {"type", "frozen", "network_params", "model_state"}, checkpoint.keys()
        network_params = cls.NetworkParams(**checkpoint["network_params"])
        assert checkpoint["type"] == cls.__name__, checkpoint["type"]
        model = initialize_model(copy.deepcopy(network_params.model))
        model.load_state_dict(checkpoint["model_state"])
        if runtime:
            network_params.runtime.update(runtime)
        return cls(model, network_params, device=device, frozen=checkpoint["frozen"])


class CirNetwork(SingleNetwork):
    def __init__(self, model, network_params, device, frozen):
        data = network_params.runtime.setdefault("data", {})
        if "mean_std" not in data:
            data["mean_std"] = [model.meta["mean"], model.meta["std"]]
        super().__init__(model, network_params, device, frozen)


NETWORKS = {"SingleNetwork": SingleNetwork, "CirNetwork": CirNetwork}


def load_checkpoint(path):
    """``Checkpoints.load_network`` for a single file (mdir/learning/checkpoints.py:145-155)."""
    from .scenario import open_resource
    checkpoint = torch.load(open_resource(path), map_location="cpu", weights_only=False)
    assert "net" not in checkpoint.get("_networks_included", {})
    return {"net": checkpoint, **checkpoint.pop("_networks_included", {})}


def initialize_network(params, device, state=None, runtime=None):
    assert state is not None, "only checkpoint-backed networks are supported on the eval path"
    kind = state["net"]["type"]
    if kind not in NETWORKS:
        raise NotImplementedError("network type %r is outside the MI355X hot path (%s are supported; "
                                  "SequentialNetwork = U-Net generator + embedding network, DESIGN.md section 7)"
                                  % (kind, sorted(NETWORKS)))
    cls = NETWORKS[kind]
    return cls.initialize_from_state({"net": state["net"]}, device, params, runtime)


def load_network(params, device):
    """``mdir/learning/__init__.py:9-11``: ``{"path": <.pth>, "runtime": <overrides>}``."""
    return initialize_network(None, device, load_checkpoint(params["path"]), params["runtime"])
