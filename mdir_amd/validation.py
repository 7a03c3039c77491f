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
    decisive_criterion = ""

    def validations(self, _epoch):
        return []

    def should_validate(self, _epoch):
        return False


class SingleValidation:
    def __init__(self, data_loader, criterion, network_overlay, frequency):
        assert data_loader is None, "only score validations (data: null) are on the MI355X path"
        self.data_loader = None
        self.criterion = criterion
        self.network_overlay = network_overlay
        self.frequency = frequency
        self.decisive_criterion = "val/learning/score:total"

    @classmethod
    def initialize(cls, params_validation, data, params_data, default_criterion, net_defaults):
        data_key = params_validation.pop("data")
        if data_key is not None:
            raise NotImplementedError("validation over a data loader is training-side and out of scope")
        criterion_section = params_validation.pop("criterion")
        if criterion_section == "default":
            if default_criterion is None:
                raise ValueError("Criterion cannot be 'default' when default criterion is not specified")
            criterion = default_criterion
        else:
            criterion = initialize_score(get_dataset_params(criterion_section, net_defaults))
        network_overlay = params_validation.pop("network_overlay")
        frequency = params_validation.pop("frequency")
        assert not params_validation, params_validation.keys()
        return cls(None, criterion, network_overlay, frequency)

    def validations(self, epoch):
        return [("val", self)] if self.should_validate(epoch) else []

    def should_validate(self, epoch):
        return epoch is None or (self.frequency and (epoch + 1) % self.frequency == 0)

    def validate(self, network, device, logger):
        network = network.overlay_params(copy.deepcopy(self.network_overlay), device)
        network.eval()
        return self.criterion(network, device, logger)


class MultiCriterialValidation:
    def __init__(self, decisive_criterion, validations):
        self.decisive_criterion = decisive_criterion
        self.vals = validations

    @classmethod
    def initialize(cls, params_validation, **kwargs):
        decisive_criterion = params_validation.pop("decisive_criterion")
        return cls(decisive_criterion, {key: initialize_validation(scenario, **kwargs)
                                        for key, scenario in params_validation.items()})

    def validations(self, epoch):
        return [(key, val) for key, val in self.vals.items() if val.should_validate(epoch)]


VALIDATIONS = {"SingleValidation": SingleValidation, "MultiCriterialValidation": MultiCriterialValidation}


def initialize_validation(params, **kwargs):
    if isinstance(params, bool) and not params:
        return NoValidation()
    return VALIDATIONS[params.pop("type")].initialize(params, **kwargs)
