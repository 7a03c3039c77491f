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
