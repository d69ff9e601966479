"""The reference's deep import paths as aliases.

normflows spreads the hot-path classes over sub-packages (`normflows.flows.affine.coupling`, `normflows.flows.neural_spline.wrapper`,
`normflows.nets.resnet`, `normflows.distributions.base`, ...); this package keeps them in fewer files.  So that
`from normflows.flows.neural_spline.coupling import PiecewiseRationalQuadraticCoupling` survives the switch of the package name,
every such dotted path of the files SURVEY.md section 8 lists resolves to an alias module that re-exports the classes it holds in
the reference (only names this package implements).  Aliases are registered in sys.modules and as attributes of their parents;
they hold no code."""
import sys
import types

# reference path (below the package) -> (module of this package that holds the names, names the reference defines there)
_TABLE = {
    "flows.affine.coupling": ("flows.affine", ["AffineConstFlow", "CCAffineConst", "AffineCoupling", "MaskedAffineFlow", "AffineCouplingBlock"]),
    "flows.affine.glow": ("flows.glow", ["GlowBlock"]),
    "flows.affine.autoregressive": ("flows.autoregressive", ["Autoregressive", "MaskedAffineAutoregressive"]),
    "flows.neural_spline.coupling": ("flows.neural_spline", ["PiecewiseRationalQuadraticCDF", "PiecewiseRationalQuadraticCoupling"]),
    "flows.neural_spline.wrapper": ("flows", ["CoupledRationalQuadraticSpline", "CircularCoupledRationalQuadraticSpline",
                                              "AutoregressiveRationalQuadraticSpline", "CircularAutoregressiveRationalQuadraticSpline"]),
    "flows.neural_spline.autoregressive": ("flows", ["MaskedPiecewiseRationalQuadraticAutoregressive"]),
    "nets.resnet": ("nets", ["ResidualBlock", "ResidualNet"]),
    "nets.mlp": ("nets", ["MLP"]),
    "nets.cnn": ("nets", ["ConvNet2d"]),
    "nets.made": ("nets", ["MaskedLinear", "MaskedFeedforwardBlock", "MaskedResidualBlock", "MADE"]),
    "distributions.base": ("distributions", ["BaseDistribution", "DiagGaussian", "ConditionalDiagGaussian", "ClassCondDiagGaussian", "GlowBase"]),
}


def install(pkg):
    """Create the alias modules below the package module `pkg` (idempotent)."""
    root = pkg.__name__
    for path, (src, names) in _TABLE.items():
        full = root + "." + path
        if full in sys.modules:
            continue
        holder = pkg
        for part in src.split("."):
            holder = getattr(holder, part)
        alias = types.ModuleType(full, "alias of the reference's module path normflows.%s (see _refpaths.py)" % path)
        exported = []
        for n in names:
            if hasattr(holder, n):
                setattr(alias, n, getattr(holder, n))
                exported.append(n)
        alias.__all__ = exported
        sys.modules[full] = alias
        parent = pkg
        parts = path.split(".")
        for part in parts[:-1]:
            parent = getattr(parent, part)
        if not hasattr(parent, parts[-1]):
            setattr(parent, parts[-1], alias)
