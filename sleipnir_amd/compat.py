"""Run a program written for the reference's Python package unchanged:

    import sleipnir_amd.compat; sleipnir_amd.compat.install()
    from sleipnir.autodiff import VariableMatrix          # -> sleipnir_amd.autodiff
    from sleipnir.optimization import Problem, bounds     # -> sleipnir_amd.optimization

install() registers the reference's module names (python/src/sleipnir/__init__.py,
autodiff/__init__.py, optimization/__init__.py) in sys.modules; it refuses to shadow an
installed `sleipnir` unless told to."""
import importlib.util
import sys
import types


def install(force: bool = False) -> None:
    if "sleipnir" in sys.modules and getattr(sys.modules["sleipnir"], "__slpx_compat__", False):
        return
    if not force and importlib.util.find_spec("sleipnir") is not None:
        raise ImportError("a `sleipnir` package is installed; pass force=True to shadow it")
    from sleipnir_amd import autodiff, optimization

    pkg = types.ModuleType("sleipnir")
    pkg.__slpx_compat__ = True
    pkg.__path__ = []  # a package: `import sleipnir.autodiff` resolves through sys.modules
    pkg.autodiff = autodiff
    pkg.optimization = optimization
    sys.modules["sleipnir"] = pkg
    sys.modules["sleipnir.autodiff"] = autodiff
    sys.modules["sleipnir.optimization"] = optimization
