"""TEST INFRASTRUCTURE ONLY -- import shim that lets the *unmodified* reference
(/root/reference, agi-brain/xuance v1.4.4) be imported in the build container.

It is used by exactly one consumer: ``oracle/make_golden.py`` (the script that
generates the committed fixtures under ``tests/golden/``).  Nothing in the
product path (``xuance_amd/``), in the ``-m gpu`` tests, in ``bench.py`` or in
``__graft_entry__.smoke()`` imports this module: /root/reference does not
exist on the GPU box.

What it stubs: ``gymnasium`` (+``.spaces``), ``wandb``,
``torch.utils.tensorboard`` and a list of optional third-party packages the
reference imports eagerly; plus one ``typing._type_check`` patch because the
reference annotates ``Optional[torch.distributions]`` which Python 3.10
rejects (xuance/torch/rl_models/modules/outputs.py:24,50,78).
"""
import sys
import types
import typing
import importlib.abc
import importlib.machinery

REFERENCE_ROOT = "/root/reference"
_INSTALLED = False


def install():
    """Install the stubs and put the reference on sys.path (idempotent)."""
    global _INSTALLED
    if _INSTALLED:
        return
    _INSTALLED = True
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class Space:
        def __init__(self, shape=None, dtype=None):
            self.shape, self.dtype = shape, dtype

        def __class_getitem__(cls, item):
            return cls

    class Box(Space):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            super().__init__(tuple(shape), dtype)
            self.low, self.high = low, high

    class Discrete(Space):
        def __init__(self, n):
            super().__init__((), None)
            self.n = n

    class Dict(Space, dict):
        def __init__(self, spaces=None):
            dict.__init__(self, spaces or {})
            self.spaces = spaces or {}

    class Tuple(Space):
        def __init__(self, spaces):
            self.spaces = spaces

    class Wrapper:
        def __init__(self, env):
            self.env = env

    sp = stub("gymnasium.spaces", Space=Space, Box=Box, Discrete=Discrete, Dict=Dict, Tuple=Tuple,
              MultiDiscrete=Space)
    stub("gymnasium", spaces=sp, Space=Space, Wrapper=Wrapper, Env=object, make=lambda *a, **k: None)
    stub("wandb")
    stub("torch.utils.tensorboard", SummaryWriter=object)

    _tc = typing._type_check
    typing._type_check = (lambda a, m, *r, **k:
                          a if isinstance(a, types.ModuleType) else _tc(a, m, *r, **k))

    optional = {"pyglet", "pygame", "cv2", "pettingzoo", "mpi4py", "smac", "torchvision", "imageio",
                "moviepy", "optuna", "ale_py", "minigrid", "d4rl", "h5py", "torch_scatter", "plotly",
                "gym", "metadrive", "rware", "gfootball", "gym_pybullet_drones", "supersuit"}

    class AutoStub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in optional:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            m = types.ModuleType(spec.name)
            m.__path__ = []

            def _getattr(a):
                if a.startswith("__"):
                    raise AttributeError(a)
                return type(a, (), {})
            m.__getattr__ = _getattr
            return m

        def exec_module(self, module):
            pass

    sys.meta_path.append(AutoStub())


def spaces():
    install()
    return sys.modules["gymnasium.spaces"]
