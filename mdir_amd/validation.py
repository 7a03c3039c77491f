"""Validation tree of a scenario -- ``mdir/learning/validation.py`` for the score
(``data: null``) branch that ``eval.py`` uses: ``SingleValidation`` (:21-76),
``MultiCriterialValidation`` (:111-141), ``VALIDATIONS`` / ``initialize_validation``
(:144-153).  Loss-criterion validations over data loaders belong to training and are
out of scope."""
import copy

from .score import initialize_score


def get_dataset_params(params, net_defaults):
    """Criterion section on top of the network's data defaults (mdir/tools/utils.py:10-11)."""
    return copy.deepcopy({**net_defaults, **params})


class NoValidation:
    """``validation: false`` (or a disabled entry of a multi-criterial tree)."""
    decisive_criterion = ""
    validations = staticmethod(lambda _epoch: [])
    should_validate = staticmethod(lambda _epoch: False)


class SingleValidation:
    """One score on one test set.  ``frequency`` only matters during training (every k-th epoch);
    ``epoch=None`` -- the stand-alone ``validate`` stage -- always runs."""

    decisive_criterion = "val/learning/score:total"

    def __init__(self, data_loader, criterion, network_overlay, frequency):
        assert data_loader is None, "only score validations (data: null) are on the MI355X path"
        self.data_loader, self.criterion = None, criterion
        self.network_overlay, self.frequency = network_overlay, frequency

    @classmethod
    def initialize(cls, params_validation, data, params_data, default_criterion, net_defaults):
        section = {key: params_validation.pop(key) for key in ("data", "criterion", "network_overlay", "frequency")}
        assert not params_validation, params_validation.keys()
        if section["data"] is not None:
            raise NotImplementedError("validation over a data loader is training-side and out of scope")
        if section["criterion"] != "default":
            criterion = initialize_score(get_dataset_params(section["criterion"], net_defaults))
        elif default_criterion is not None:
            criterion = default_criterion
        else:
            raise ValueError("Criterion cannot be 'default' when default criterion is not specified")
        return cls(None, criterion, section["network_overlay"], section["frequency"])

    def should_validate(self, epoch):
        if epoch is None:
            return True
        return bool(self.frequency) and (epoch + 1) % self.frequency == 0

    def validations(self, epoch):
        return [("val", self)] if self.should_validate(epoch) else []

    def validate(self, network, device, logger):
        evaluated = network.overlay_params(copy.deepcopy(self.network_overlay), device)
        evaluated.eval()
        return self.criterion(evaluated, device, logger)


class MultiCriterialValidation:
    """Named sub-validations; ``validations(epoch)`` yields the ones that are due."""

    def __init__(self, decisive_criterion, validations):
        self.decisive_criterion, self.vals = decisive_criterion, validations

    @classmethod
    def initialize(cls, params_validation, **kwargs):
        decisive = params_validation.pop("decisive_criterion")
        children = {name: initialize_validation(sub, **kwargs) for name, sub in params_validation.items()}
        return cls(decisive, children)

    def validations(self, epoch):
        return [(name, child) for name, child in self.vals.items() if child.should_validate(epoch)]


VALIDATIONS = {"SingleValidation": SingleValidation, "MultiCriterialValidation": MultiCriterialValidation}


def initialize_validation(params, **kwargs):
    if params is False:
        return NoValidation()
    return VALIDATIONS[params.pop("type")].initialize(params, **kwargs)
