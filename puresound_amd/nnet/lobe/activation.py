"""Activation lookup for the U-Net family: which nn class a recipe's `activation_type` string names (so the parameter
tree and state_dict keys match the reference's lobe/activation.py) and which ps_activation_f32 kind computes it."""
import torch.nn as nn

_TABLE = {"relu": nn.ReLU, "prelu": nn.PReLU, "mish": nn.Mish, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh}


def get_activation(name: str):
    """Class for a lower-case activation name; anything else is a NameError, as in the reference."""
    try:
        return _TABLE[name]
    except (KeyError, TypeError):
        raise NameError("Could not interpret activation identifier") from None


def activation_kind(mod: nn.Module) -> str:
    """ps_activation_f32 kind of an instantiated activation module."""
    for kind, cls in _TABLE.items():
        if isinstance(mod, cls):
            return kind
    raise NotImplementedError(type(mod).__name__)
