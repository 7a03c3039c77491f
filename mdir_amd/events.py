"""Metric bookkeeping of the eval surface: the part of mdir's event system that turns
logger rows into the numbers ``eval.py`` prints.

``EventBroker.register_data`` / ``close_epoch`` / ``metadata.metadata()`` follow
``mdir/tools/eventprocessor.py:643-669`` and ``MetadataKeeper`` (:54-121): rows that
arrive with an iteration index are collected into per-key lists, rows without one
are scalars; at epoch close every ``scalar/loss|score`` list becomes its nan-filtered
mean under ``"<key>:<subkey>_avg.4"``, ``scalar/time`` lists their sum under
``"_sum.1"``, scalars keep ``"<key>:<subkey>"``.  Tensorboard / HTML sinks are
reporting and out of scope.
"""
import numpy as np

_SUFFIX = {"avg": "_avg.4", "sum": "_sum.1", None: ""}


class MetadataKeeper:
    def __init__(self):
        self.data = {}

    def register_epoch_data(self, epoch, rows):
        for key, item in rows.items():
            if not item["dtype"].startswith("scalar/"):
                continue
            for subkey, value in item["data"].items():
                if not isinstance(value, (list, np.ndarray)):
                    aggr = None
                else:
                    aggr = "avg" if item["dtype"] in {"scalar/loss", "scalar/score"} else "sum"
                slot = self.data.setdefault((key, subkey), {"dtype": item["dtype"], "aggr": aggr,
                                                            "key": key + ":" + subkey + _SUFFIX[aggr],
                                                            "epochs": [], "data": []})
                v = np.array(value)
                if aggr is not None:
                    v = v[~np.isnan(v)]
                    v = {"avg": np.mean, "sum": np.sum}[aggr](v)
                slot["epochs"].append(epoch)
                slot["data"].append(v)

    def metadata(self):
        return {y["key"]: y["data"] for y in self.data.values() if y["dtype"] in {"scalar/loss", "scalar/score"}}


class EventBroker:
    def __init__(self):
        self.metadata = MetadataKeeper()
        self.epoch = 0
        self._rows = {}

    def register_data(self, epoch, relative_iteration, epoch_size, key, data, dtype):
        slot = self._rows.get(key)
        if slot is None:
            slot = self._rows[key] = {"dtype": dtype, "epoch_size": epoch_size, "relative_iteration": None,
                                      "data": {}}
        assert slot["dtype"] == dtype, (key, slot["dtype"], dtype)
        if relative_iteration is None:
            assert not slot["data"], "scalar row '%s' registered twice" % key
            slot["data"] = dict(data)
        else:
            if slot["relative_iteration"] is None:
                slot["relative_iteration"] = []
                slot["data"] = {k: [] for k in data}
            assert slot["data"].keys() == data.keys(), (slot["data"].keys(), data.keys())
            slot["relative_iteration"].append(relative_iteration)
            for k, v in data.items():
                slot["data"][k].append(v)

    def close_epoch(self):
        self.metadata.register_epoch_data(self.epoch, self._rows)
        self._rows = {}
        self.epoch += 1


def initialize_processor(params=None, dataroot=None):
    """Signature-compatible stand-in for ``initialize_processor`` (eventprocessor.py:693-697);
    streamer options such as ``progress`` are accepted and ignored."""
    return EventBroker()
