"""Activation lookup (mirror of puresound/nnet/lobe/activation.py): the names the recipes pass and the classes that
hold their parameters; the arithmetic is ps_activation_f32."""
import torch.nn as nn

relu = nn.ReLU
prelu = nn.PReLU
mish = nn.Mish
sigmoid = nn.Sigmoid
tanh = nn.Tanh


def get_activation(name: str):
    if name not in ["relu", "mish", "prelu", "sigmoid", "tanh"]:
        raise NameError("Could not interpret activation identifier")
    return globals()[name]


def activation_kind(mod: nn.Module) -> str:
    for kind, cls in (("relu", nn.ReLU), ("prelu", nn.PReLU), ("mish", nn.Mish), ("sigmoid", nn.Sigmoid),
                      ("tanh", nn.Tanh)):
        if isinstance(mod, cls):
            return kind
    raise NotImplementedError(type(mod).__name__)
