"""Conditioner networks (L1 of the reference): ResidualNet / ResidualBlock, MLP, ConvNet2d.

Mirrors normflows/nets/resnet.py:7-104, nets/mlp.py:5-58, nets/cnn.py:5-63: constructor signatures, parameter
registration and RNG consumption order (a seeded build reproduces the reference's initial weights) and
state_dict keys.  These modules are the MFMA-shaped part of the path.  `forward` here issues plain library
GEMMs / convolutions through torch (rocBLAS / hipBLASLt / MIOpen); the NSF coupling layer bypasses
ResidualNet.forward and feeds the same weights to the hand-written fused MFMA kernel when the shape allows.
"""
import torch
from torch import nn
from torch.nn import functional as F, init


class ResidualBlock(nn.Module):
    """Pre-activation residual block x + W2 act(W1 act(x)) with optional GLU context gate (resnet.py:7-50)."""

    def __init__(self, features, context_features, activation=F.relu, dropout_probability=0.0, use_batch_norm=False,
                 zero_initialization=True):
        super().__init__()
        self.activation = activation
        self.use_batch_norm = use_batch_norm
        if use_batch_norm:
            self.batch_norm_layers = nn.ModuleList([nn.BatchNorm1d(features, eps=1e-3) for _ in range(2)])
        if context_features is not None:
            self.context_layer = nn.Linear(context_features, features)
        self.linear_layers = nn.ModuleList([nn.Linear(features, features) for _ in range(2)])
        self.dropout = nn.Dropout(p=dropout_probability)
        if zero_initialization:
            init.uniform_(self.linear_layers[-1].weight, -1e-3, 1e-3)
            init.uniform_(self.linear_layers[-1].bias, -1e-3, 1e-3)

    def forward(self, inputs, context=None):
        temps = inputs
        if self.use_batch_norm:
            temps = self.batch_norm_layers[0](temps)
        temps = self.linear_layers[0](self.activation(temps))
        if self.use_batch_norm:
            temps = self.batch_norm_layers[1](temps)
        temps = self.linear_layers[1](self.dropout(self.activation(temps)))
        if context is not None:
            temps = F.glu(torch.cat((temps, self.context_layer(context)), dim=1), dim=1)
        return inputs + temps


class ResidualNet(nn.Module):
    """initial Linear -> num_blocks ResidualBlocks -> final Linear (resnet.py:53-104)."""

    def __init__(self, in_features, out_features, hidden_features, context_features=None, num_blocks=2,
                 activation=F.relu, dropout_probability=0.0, use_batch_norm=False, preprocessing=None):
        super().__init__()
        self.hidden_features = hidden_features
        self.context_features = context_features
        self.preprocessing = preprocessing
        if context_features is not None:
            self.initial_layer = nn.Linear(in_features + context_features, hidden_features)
        else:
            self.initial_layer = nn.Linear(in_features, hidden_features)
        self.blocks = nn.ModuleList([
            ResidualBlock(features=hidden_features, context_features=context_features, activation=activation,
                          dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
            for _ in range(num_blocks)
        ])
        self.final_layer = nn.Linear(hidden_features, out_features)
        self.dropout_probability = dropout_probability
        self.use_batch_norm = use_batch_norm

    def forward(self, inputs, context=None):
        temps = inputs if self.preprocessing is None else self.preprocessing(inputs)
        if context is None:
            temps = self.initial_layer(temps)
        else:
            temps = self.initial_layer(torch.cat((temps, context), dim=1))
        for block in self.blocks:
            temps = block(temps, context=context)
        return self.final_layer(temps)

    def is_plain_relu(self):
        """True when the net is the plain ReLU MLP the fused HIP kernel implements."""
        act_ok = all(isinstance(b.activation, nn.ReLU) or b.activation is F.relu for b in self.blocks)
        return (act_ok and self.context_features is None and self.preprocessing is None and not self.use_batch_norm
                and (self.dropout_probability == 0.0 or not self.training))


class ConstScaleLayer(nn.Module):
    """utils/nn.py ConstScaleLayer: multiply by a constant."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale_cpu = torch.tensor(scale)
        self.register_buffer("scale", self.scale_cpu)

    def forward(self, input):
        return input * self.scale


class ClampExp(nn.Module):
    """utils/nn.py ClampExp: exp(min(x, 1))."""

    def forward(self, x):
        one = torch.tensor(1.0, device=x.device, dtype=x.dtype)
        return torch.exp(torch.min(x, one))


class MLP(nn.Module):
    """Linear + LeakyReLU stack (mlp.py:5-58)."""

    def __init__(self, layers, leaky=0.0, score_scale=None, output_fn=None, output_scale=None, init_zeros=False,
                 dropout=None):
        super().__init__()
        net = nn.ModuleList([])
        for k in range(len(layers) - 2):
            net.append(nn.Linear(layers[k], layers[k + 1]))
            net.append(nn.LeakyReLU(leaky))
        if dropout is not None:
            net.append(nn.Dropout(p=dropout))
        net.append(nn.Linear(layers[-2], layers[-1]))
        if init_zeros:
            nn.init.zeros_(net[-1].weight)
            nn.init.zeros_(net[-1].bias)
        if output_fn is not None:
            if score_scale is not None:
                net.append(ConstScaleLayer(score_scale))
            if output_fn == "sigmoid":
                net.append(nn.Sigmoid())
            elif output_fn == "relu":
                net.append(nn.ReLU())
            elif output_fn == "tanh":
                net.append(nn.Tanh())
            elif output_fn == "clampexp":
                net.append(ClampExp())
            else:
                raise NotImplementedError("This output function is not implemented.")
            if output_scale is not None:
                net.append(ConstScaleLayer(output_scale))
        self.net = nn.Sequential(*net)

    def forward(self, x):
        return self.net(x)


class _NetActNorm(nn.Module):
    """utils/nn.py ActNorm wrapper: the flow's forward output without the log-det."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        from .flows.normalization import ActNorm
        self.actNorm = ActNorm(*args, **kwargs)

    def forward(self, input):
        out, _ = self.actNorm(input)
        return out


class ConvNet2d(nn.Module):
    """Conv2d + LeakyReLU stack used by GlowBlock (cnn.py:5-63)."""

    def __init__(self, channels, kernel_size, leaky=0.0, init_zeros=True, actnorm=False, weight_std=None):
        super().__init__()
        net = nn.ModuleList([])
        for i in range(len(kernel_size) - 1):
            conv = nn.Conv2d(channels[i], channels[i + 1], kernel_size[i], padding=kernel_size[i] // 2,
                             bias=(not actnorm))
            if weight_std is not None:
                conv.weight.data.normal_(mean=0.0, std=weight_std)
            net.append(conv)
            if actnorm:
                net.append(_NetActNorm((channels[i + 1],) + (1, 1)))
            net.append(nn.LeakyReLU(leaky))
        i = len(kernel_size)
        net.append(nn.Conv2d(channels[i - 1], channels[i], kernel_size[i - 1], padding=kernel_size[i - 1] // 2))
        if init_zeros:
            nn.init.zeros_(net[-1].weight)
            nn.init.zeros_(net[-1].bias)
        self.net = nn.Sequential(*net)

    def forward(self, x):
        return self.net(x)
